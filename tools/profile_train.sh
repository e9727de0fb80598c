cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r1t
mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats -d $O/stats -o bench -- python3 bench.py --mode ${1:-train} --steps 6 --warmup 2 > $O/bench_stats.log 2>&1
tail -1 $O/bench_stats.log | cut -c1-200
python3 - <<'PY'
import sqlite3, os
c=sqlite3.connect(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r1t/stats/bench_results.db')
tot=c.execute("select sum(total_duration) from top_kernels").fetchone()[0]
print("total kernel time per step: %.2f ms" % (tot/8/1000))
for r in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 34"):
    print("%-86s %5d %9.1f %8.1f %6.2f" % (r[0][:86], r[1]//8, r[2]/8, r[3], r[4]))
PY
