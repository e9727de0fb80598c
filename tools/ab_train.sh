#!/bin/bash
# run on the GPU box: image / video training step (replayed graph) under settings of one environment variable, interleaved
# tools/ab_train.sh <VAR> "<values>" <reps> [train|train_video]
VAR=$1; VALS=$2; REPS=${3:-2}; MODE=${4:-train}
for rep in $(seq 1 $REPS); do
  for v in $VALS; do
    env $VAR=$v UNCL_BENCH_WGRAD=0 python bench.py --mode $MODE --no-eager --steps 30 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$MODE $VAR=$v', round(d['ms_per_step'],3), d.get('ms_per_step_median'))"
  done
done
