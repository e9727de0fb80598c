#!/bin/bash
# run on the GPU box: whole-forward ms per step under settings of one environment variable, interleaved
# tools/ab_fwd.sh <VAR> "<values>" <reps>
VAR=$1; VALS=$2; REPS=${3:-3}
for rep in $(seq 1 $REPS); do
  for v in $VALS; do
    env $VAR=$v python bench.py --no-train --no-cpu --no-layers --no-exclusive --no-4k --sustain-seconds 0 --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', round(d['ms_per_step'],3))"
  done
done
