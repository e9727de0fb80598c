#!/bin/bash
# run on the GPU box: LDS bank-conflict share per kernel of the eager image training step -> gpurun_out/<tag>/train_lds.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r4}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export UNCL_TRAIN_GRAPH=0 UNCL_BWD_WSTREAM=0 UNCL_BENCH_MEMMAP=0
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_LDS -d $O/tl -o bench -- python3 bench.py --mode ${2:-train} --no-eager --steps 4 --warmup 2 > $O/tl.log 2>&1
python3 - <<PY
import sys
sys.path.insert(0, "$R/tools")
import pmc_summary as P
c = P._demangled(P.read("$O/tl", "SQ_LDS_BANK_CONFLICT"))
a = P._demangled(P.read("$O/tl", "SQ_LDS_IDX_ACTIVE"))
rows = []
for k, (v, n, us) in c.items():
    act = a.get(k, [0, 1, 0])[0]
    rows.append((us, k[0][:95], k[1], n, v / max(act, 1), act / max(n, 1)))
rows.sort(reverse=True)
with open("$O/train_lds.txt", "w") as out:
    out.write("%-95s %9s %6s %10s %10s %12s\n" % ("kernel", "grid", "calls", "total us", "conflict", "LDS cyc/launch"))
    for us, name, grid, n, share, act in rows[:40]:
        out.write("%-95s %9d %6d %10.1f %10.3f %12.3g\n" % (name, grid, n, us, share, act))
PY
rm -rf $O/tl
head -32 $O/train_lds.txt
