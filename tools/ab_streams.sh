#!/bin/bash
# run on the GPU box: forward step with 1..4 stream parts (UNCL_STREAMS), interleaved
for rep in 1 2; do for n in 1 2 3 4; do
UNCL_STREAMS=$n python bench.py --no-train --no-cpu --no-layers --no-exclusive 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams=$n', round(d['value'],1), round(d['ms_per_step'],3))"
done; done
