#!/bin/bash
# run on the GPU box: memory-path / issue-stall counters of single layers (product library, producer / consumer kernel only)
#   bash tools/pmc_layers2.sh <tag> <layers>   -> gpurun_out/<tag>/pmc_layers2.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r4}; LAYERS=${2:-incf,d0b,up3f}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
PASSES=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VALU_MFMA_COEXEC_CYCLES"
        "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VALU_MFMA_BUSY_CYCLES")
# (TA_* / TCP_* counters are NOT collected: a pass with TA_BUSY_sum ... aborted rocprofv3 (signal 6) and then sat on the box until the
# caller's limit killed it: 25 GPU-minutes for nothing)
: > $O/pmc_layers2.txt
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $P -d $O/p2_$i -o pl -- python3 tools/pc_phase_timing.py --product --only-pc 2 --layers $LAYERS --reps 3 > $O/p2_$i.log 2>&1
  python3 - "$O/p2_$i" $P >> $O/pmc_layers2.txt <<'PY'
import sys
sys.path.insert(0, sys.argv[0] and ".")
sys.path.insert(0, "tools")
import pmc_summary as P
d, counters = sys.argv[1], sys.argv[2:]
per = {}
for c in counters:
    try:
        rows = P._demangled(P.read(d, c))
    except Exception as e:
        print("counter", c, "failed", e); continue
    for k, (v, n, us) in rows.items():
        if "conv3x3_pc_kernel" not in k[0]: continue
        per.setdefault(k, {})[c] = v / max(n, 1)
for k, e in per.items():
    print(k[0][-62:], " ".join("%s=%.4g" % (c.replace("_sum", ""), v) for c, v in e.items()))
PY
  rm -rf $O/p2_$i
done
cat $O/pmc_layers2.txt
