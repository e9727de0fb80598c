#!/bin/bash
# run on the GPU box: a training step with an environment switch at two values, interleaved:
#   tools/ab_train.sh VAR [reps] [mode] [value_a] [value_b]
VAR=${1:-UNCL_WG_WIDE}; REPS=${2:-2}; MODE=${3:-train}; A=${4:-0}; B=${5:-1}
mkdir -p gpurun_out/ab_train
for rep in $(seq 1 $REPS); do
for v in $A $B; do
  env $VAR=$v python bench.py --mode $MODE --steps 30 --warmup 5 > gpurun_out/ab_train/out.json 2> gpurun_out/ab_train/err_${v}_$rep.log
  rc=$?
  if [ $rc -ne 0 ]; then echo "$VAR=$v rc=$rc"; grep -v "amdgpu.ids\|Warning\|run_backward" gpurun_out/ab_train/err_${v}_$rep.log | tail -3; continue; fi
  python -c "import sys,json; d=json.loads(open('gpurun_out/ab_train/out.json').read().strip().splitlines()[-1]); print('$VAR=$v', 'median', round(d['ms_median'],3), 'min', round(d['ms_min'],3))"
done; done
