#!/bin/bash
# run on the GPU box: SQ counters of single layers (product library, producer / consumer kernel only)
#   bash tools/pmc_layers.sh <tag> <layers>   -> gpurun_out/<tag>/pmc_layers.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r4}; LAYERS=${2:-incf,d0b,up3f}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/pl -o pl -- python3 tools/pc_phase_timing.py --product --only-pc 2 --layers $LAYERS --reps 3 > $O/pl.log 2>&1
python3 - <<PY
import sys
sys.path.insert(0, "$R/tools")
import pmc_summary as P
C = ["SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"]
per = {}
for c in C:
    for k, (v, n, us) in P._demangled(P.read("$O/pl", c)).items():
        if "conv3x3_pc_kernel" not in k[0]: continue
        e = per.setdefault(k, {"n": n, "us": us}); e[c] = v / max(n, 1)
with open("$O/pmc_layers.txt", "w") as out:
    for k, e in per.items():
        out.write("%s grid %d launches %d avg %.1f us\n" % (k[0][-70:], k[1], e["n"], e["us"] / e["n"]))
        out.write("   " + "  ".join("%s %.3g" % (c[3:], e.get(c, 0)) for c in C) + "\n")
        if e.get("SQ_INSTS_MFMA"): out.write("   VALU/MFMA %.2f  LDS/MFMA %.2f  bank-conflict share of LDS cycles %.3f  WAIT_ANY share %.3f  WAIT_INST share %.3f\n" % (
            e["SQ_INSTS_VALU"] / e["SQ_INSTS_MFMA"], e["SQ_INSTS_LDS"] / e["SQ_INSTS_MFMA"], e["SQ_LDS_BANK_CONFLICT"] / max(e["SQ_LDS_IDX_ACTIVE"], 1),
            e["SQ_WAIT_ANY"] / e["SQ_WAVE_CYCLES"], e["SQ_WAIT_INST_ANY"] / e["SQ_WAVE_CYCLES"]))
PY
rm -rf $O/pl
cat $O/pmc_layers.txt
