#!/bin/bash
# run on the GPU box: the training step of this tree against a second tree (an older commit unpacked under tools/_ab/base_tree (git archive <commit> | tar -x -C tools/_ab/base_tree) with the
# current library copied in), interleaved.   tools/ab_tree.sh <other tree> <reps> [train|train_video]
OTHER=$1; REPS=${2:-2}; MODE=${3:-train}
for rep in $(seq 1 $REPS); do
  for t in . $OTHER; do
    (cd $t && UNCL_BENCH_WGRAD=0 python bench.py --mode $MODE --no-eager --steps 30 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$MODE tree=$t', round(d['ms_per_step'],3))")
  done
done
