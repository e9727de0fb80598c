#!/bin/bash
# issue counters of the Gaussian-statistics kernels (tools/gauss_ab.py 8): what bounds the backward's separable passes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-pmcg}; mkdir -p $O; cd $R
for P in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  rocprofv3 --kernel-trace --pmc $P -d $O/p -o g -- python3 tools/gauss_ab.py 8 > $O/p.log 2>&1
  python3 - "$O/p" $P <<'PY'
import sys
sys.path.insert(0, sys.argv[0] and "tools" or "tools")
sys.path.insert(0, "tools")
import pmc_summary as P
d = sys.argv[1]
for cn in sys.argv[2:]:
    try:
        rows = P._demangled(P.read(d, cn))
    except Exception as e:
        print(cn, "ERR", e); continue
    for k, (v, n, us) in rows.items():
        if "gauss" in k[0]:
            print("%-24s %-60s calls %4d  per launch %.4g   us/launch %.1f" % (cn, k[0][:60], n, v / max(n, 1), us / max(n, 1)))
PY
  rm -rf $O/p
done
