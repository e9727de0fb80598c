"""Per-launch durations of the weight-gradient kernels of ONE training step, in launch order, from a rocprofv3 kernel trace:
  rocprofv3 --kernel-trace -d gpurun_out/X/wtrace -o t -- python3 bench.py --mode train --steps 3 --warmup 1
  python3 tools/wgrad_trace.py gpurun_out/X/wtrace [pattern]
The backward pass launches them from the last decoder layer to the first encoder layer (generator.hip: uncl_gen_backward)."""
import glob
import os
import sqlite3
import sys

base = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "wgrad"
rows = []
for f in glob.glob(os.path.join(base, "**", "*_results.db"), recursive=True):
    db = sqlite3.connect(f)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    for r in db.execute("select S.kernel_name, K.grid_size_x, K.grid_size_y, K.grid_size_z, K.workgroup_size_x, K.start, K.end "
                        "from %s K inner join %s S on S.id = K.kernel_id and S.guid = K.guid order by K.start" % (kd, ks)):
        rows.append(r)
sel = [r for r in rows if pat in r[0]]
if not sel:
    print("no kernel matches", pat, "of", len(rows))
    sys.exit(0)
# the last step: launches after the last big gap are the tail; simply take the last third of the matches
per_step = len(sel) // max(1, int(os.environ.get("STEPS", "4")))
tail = sel[-per_step:]
tot = 0.0
for name, gx, gy, gz, wx, st, en in tail:
    short = name.split("(")[0][-40:]
    us = (en - st) / 1e3
    tot += us
    print("%-42s grid %7d x %2d x %4d (wg %4d)  %8.1f us" % (short, gx // max(wx, 1), gy, gz, wx, us))
print("launches %d, total %.1f us" % (len(tail), tot))
