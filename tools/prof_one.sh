#!/bin/bash
# run on the GPU box: tools/prof_one.sh <tag> <name> <bench.py args...>
#   rocprofv3 --kernel-trace --stats over `python3 bench.py <args>` -> gpurun_out/<tag>/<name>_kernel_stats.csv (+ the line it printed)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; NAME=$2; shift 2
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats -d $O/db_$NAME -o bench -- python3 bench.py "$@" > $O/${NAME}_under_rocprof.log 2>&1
python3 - <<PY
import sys
sys.path.insert(0, "$R/tools")
import stats_summary
stats_summary.dump("$O/db_$NAME", "$O/${NAME}_kernel_stats.csv")
PY
rm -rf $O/db_$NAME
grep -v amdgpu.ids $O/${NAME}_under_rocprof.log | tail -1 | cut -c1-400
