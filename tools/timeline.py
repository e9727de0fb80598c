"""One step's kernel timeline from a rocprofv3 --kernel-trace database: start offset, duration, queue, kernel (short name).
Run on the GPU box (cd /tmp first):
    rocprofv3 --kernel-trace -d gpurun_out/tl -o bench -- python3 bench.py --steps 6 --warmup 2 --no-cpu --no-train --no-exclusive --no-layers --no-4k --sustain-seconds 0
    python3 tools/timeline.py gpurun_out/tl [step_index_from_the_end] [marker kernel substring: once per step, first]"""
import glob
import os
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", n)
    m = re.match(r"(conv3x3_\w+?_kernel)I(.*?)EEv8PipeArgs", n)
    if m:
        args = m.group(2).replace("DF16b", "bf16,").replace("DF16_", "f16,").replace("Li", "").replace("Lb", "").replace("E", ",")
        return m.group(1) + "<" + args.strip(",") + ">"
    return n[:60]


def main():
    d = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    marker = sys.argv[3] if len(sys.argv) > 3 else "tile_gather"      # a kernel that runs once per step, first
    f = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)[0]
    c = sqlite3.connect(f)
    rows = c.execute("select S.kernel_name, K.start, K.end, K.queue_id from rocpd_kernel_dispatch K inner join rocpd_info_kernel_symbol S "
                     "on S.id = K.kernel_id and S.guid = K.guid order by K.start").fetchall()
    # a step starts at a tile_gather launch
    starts = [i for i, r in enumerate(rows) if marker in r[0]]
    i0 = starts[-back]
    i1 = starts[-back + 1] if back > 1 else len(rows)
    t0 = rows[i0][1]
    qs = sorted({r[3] for r in rows[i0:i1]})
    print("step of %d launches, %.3f ms from first start to last end" % (i1 - i0, (max(r[2] for r in rows[i0:i1]) - t0) / 1e6))
    for name, s, e, q in rows[i0:i1]:
        print("%9.1f us  +%8.1f us  q%d  %s" % ((s - t0) / 1e3, (e - s) / 1e3, qs.index(q), short(name)))


if __name__ == "__main__":
    main()
