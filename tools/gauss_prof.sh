#!/bin/bash
# kernel durations of the Gaussian-statistics kernels (tools/gauss_ab.py under rocprofv3) for a list of "VAR=value" settings
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-gauss}; mkdir -p $O; cd $R; shift
for v in "$@"; do
  export $v
  echo "== $v"
  rocprofv3 --kernel-trace --stats -d $O/t -o g -- python3 tools/gauss_ab.py ${GAUSS_N:-8} > $O/ab.log 2>&1
  python3 - $O/t <<'PY'
import sqlite3, sys, glob
db = glob.glob(sys.argv[1] + "/**/*results.db", recursive=True)[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
tk = [x for x in tabs if x.startswith("top_kernels")][0]
for r in c.execute("select name,total_calls,total_duration,average from %s" % tk):
    if "gauss" in r[0]:
        print("%-70s calls %4d avg %8.1f us" % (r[0][:70], r[1], r[3]))
PY
  rm -rf $O/t
  unset ${v%%=*}
done
