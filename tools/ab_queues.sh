run() { python bench.py --no-train --no-cpu --no-layers --no-exclusive 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['ms_per_step'],3))"; }
for i in 1 2; do
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 UNCL_FORCE_DIST=1 MASTER_PORT=2953$i
UNCL_SIDE_PRIORITY=1 run "dist=1 prio=1"
UNCL_SIDE_PRIORITY=0 run "dist=1 prio=0"
unset RANK LOCAL_RANK WORLD_SIZE MASTER_ADDR UNCL_FORCE_DIST MASTER_PORT
UNCL_SIDE_PRIORITY=1 run "dist=0 prio=1"
UNCL_SIDE_PRIORITY=0 run "dist=0 prio=0"
UNCL_SIDE_PRIORITY=1 GPU_MAX_HW_QUEUES=1 run "dist=0 prio=1 queues=1"
done
