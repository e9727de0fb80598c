#!/bin/bash
# run on the GPU box: FETCH_SIZE / WRITE_SIZE per kernel of the (eager, single-process) image training step
#   tools/pmc_train.sh <tag>  -> gpurun_out/<tag>/train_pmc.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r4}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export UNCL_TRAIN_GRAPH=0 UNCL_BWD_WSTREAM=0 UNCL_BENCH_MEMMAP=0
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/tp_fetch -o bench -- python3 bench.py --mode train --no-eager --steps 4 --warmup 2 > $O/tp_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/tp_write -o bench -- python3 bench.py --mode train --no-eager --steps 4 --warmup 2 > $O/tp_write.log 2>&1
python3 - <<PY
import sys
sys.path.insert(0, "$R/tools")
import pmc_summary as P
f = P._demangled(P.read("$O/tp_fetch", "FETCH_SIZE"))
w = P._demangled(P.read("$O/tp_write", "WRITE_SIZE"))
rows = []
for k, (fs, n, us) in f.items():
    ws = w.get(k, [0.0, 1, 0.0])
    # counters are KiB; FETCH_SIZE counts wide reads at half (MI355X_MICROARCH.md)
    rows.append((us, k[0][:100], k[1], n, 2 * fs * 1024 / n / 1e6, ws[0] * 1024 / max(ws[1], 1) / 1e6, us / n))
rows.sort(reverse=True)
with open("$O/train_pmc.txt", "w") as out:
    out.write("%-100s %9s %6s %10s %10s %9s\n" % ("kernel", "grid", "calls", "fetch MB", "write MB", "avg us"))
    for us, name, grid, n, fm, wm, avg in rows[:60]:
        out.write("%-100s %9d %6d %10.1f %10.1f %9.1f\n" % (name, grid, n, fm, wm, avg))
PY
rm -rf $O/tp_fetch $O/tp_write
head -40 $O/train_pmc.txt
