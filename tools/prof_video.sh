#!/bin/bash
# kernel trace of the replayed video training step (2 clips): per-kernel totals per step
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-vid}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
export UNCL_BENCH_WGRAD=0
STEPS=10; WARM=2
rocprofv3 --kernel-trace --stats -d $O/train_video -o bench -- python3 bench.py --mode ${MODE:-train_video} --no-eager --steps $STEPS --warmup $WARM > $O/train_video_stats.log 2>&1
tail -1 $O/train_video_stats.log | cut -c1-300
python3 - $O <<'PY'
import sqlite3, sys, glob
db = glob.glob(sys.argv[1] + "/train_video/**/*results.db", recursive=True)[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
t = [x for x in tabs if x.startswith("top_kernels")][0]
rows = list(c.execute("select name,total_calls,total_duration,average from %s" % t))
tot = sum(r[2] for r in rows); calls = sum(r[1] for r in rows)
print("all launches %d, total kernel time %.1f ms" % (calls, tot / 1e6))
with open(sys.argv[1] + "/video_kernel_stats.txt", "w") as f:
    for r in sorted(rows, key=lambda r: -r[2])[:60]:
        line = "%-110s %6d %10.1f us %8.1f us %5.2f%%" % (r[0][:110], r[1], r[2] / 1e3, r[3] / 1e3, 100.0 * r[2] / tot)
        print(line); f.write(line + "\n")
PY
rm -rf $O/train_video
