# LDS / issue counters of the producer-consumer conv kernel on one layer (run on the GPU box): tools/pmc_pc.sh up3 2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
L=${1:-up3}; PC=${2:-2}
O=$R/gpurun_out/pmc_pc_${L}_$PC
mkdir -p $O
cd $R
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  tag=$(echo $grp | cut -d" " -f1)
  rocprofv3 --kernel-trace --pmc $grp -d $O/$tag -o p -- python3 tools/pc_phase_timing.py --product --only-pc $PC --layers $L --reps 3 > $O/$tag.log 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("$O/*/*/*counter_collection.csv") + glob.glob("$O/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        if "conv3x3" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s %.4g per launch" % (c, v / max(cnt[(k, c)], 1)))
PY
