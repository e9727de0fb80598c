cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r1h
mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats -d $O/stats -o bench -- python3 bench.py --steps 10 --warmup 2 --no-cpu > $O/bench_stats.log 2>&1
python3 - <<'PY'
import sqlite3, os
c=sqlite3.connect(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r1h/stats/bench_results.db')
for r in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 18"):
    print("%-90s %4d %10.1f %8.1f %6.2f" % (r[0][:90], r[1], r[2], r[3], r[4]))
PY
