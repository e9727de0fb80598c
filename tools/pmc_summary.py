"""Summarise the two rocprofv3 PMC passes of tools/profile_bench.sh (FETCH_SIZE, WRITE_SIZE; counter-collection CSVs) into
profiles/<tag>_pmc_traffic.json: HBM bytes per launch for every kernel of the bench step, gfx950 correction applied
(MI355X_MICROARCH.md, HBM section: counters are KiB, FETCH_SIZE counts wide reads at half).

  python tools/pmc_summary.py gpurun_out/r1g r1g [steps_in_pmc_run]
"""
import collections
import csv
import glob
import json
import os
import sqlite3
import sys

DOMINANT = "conv3x3_pipe_kernel<1, 4, 4, 4, false, false>"    # up_path.3.conv.conv (loader-fused up-conv)


def read(dirname, counter):
    """(kernel name, grid) -> [sum of counter values, launches, sum of durations in us]; rocprofv3 writes either a
    counter-collection CSV or a rocpd sqlite database depending on its output format"""
    rows = collections.OrderedDict()

    def add(name, grid, value, start, end):
        e = rows.setdefault((name, int(grid)), [0.0, 0, 0.0])
        e[0] += float(value)
        e[1] += 1
        e[2] += (int(end) - int(start)) / 1e3

    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                add(r["Kernel_Name"], r["Grid_Size"], r["Counter_Value"], r["Start_Timestamp"], r["End_Timestamp"])
    for f in glob.glob(os.path.join(dirname, "**", "*_results.db"), recursive=True):
        db = sqlite3.connect(f)
        for r in db.execute("select kernel_name, grid_size, value, start, end from counters_collection where counter_name = ? "
                            "order by dispatch_id", (counter,)):
            add(*r)
    return rows


def main():
    base, tag = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5        # timed + warm-up steps of the PMC run
    fetch, write = read(os.path.join(base, "pmc_fetch"), "FETCH_SIZE"), read(os.path.join(base, "pmc_write"), "WRITE_SIZE")
    kernels, total = [], 0.0
    for key, (fs, n, us) in fetch.items():
        ws = write.get(key, [0.0, 1, 0.0])
        f_kib, w_kib = fs / n, ws[0] / max(ws[1], 1)
        per_launch = (2.0 * f_kib + w_kib) * 1024.0
        total += per_launch * n / steps
        kernels.append({"kernel": key[0], "grid_x": key[1], "launches": n, "avg_us_under_pmc": round(us / n, 1),
                        "FETCH_SIZE_KiB": round(f_kib, 1), "WRITE_SIZE_KiB": round(w_kib, 1), "hbm_bytes_per_launch": int(per_launch)})
    kernels.sort(key=lambda k: -k["avg_us_under_pmc"] * k["launches"])
    dom = next((k for k in kernels if DOMINANT in k["kernel"]), None)
    doc = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separately, WRITE_SIZE) -- python3 bench.py --steps 4 "
                      "--warmup 1 --no-cpu   --no-exclusive (tools/profile_bench.sh, tools/pmc_summary.py)",
           "correction": "gfx950: FETCH_SIZE counts wide coalesced reads at 1/2 (MI355X_MICROARCH.md, HBM section) -> hbm_bytes = "
                         "(2*FETCH_SIZE + WRITE_SIZE) * 1024; counters are KiB per launch, averaged over the run's launches",
           "dominant": {"kernel": (dom["kernel"] if dom else DOMINANT) + " @ up_path.3.conv.conv", "grid_x": dom["grid_x"] if dom else None,
                        "hbm_bytes_per_launch": dom["hbm_bytes_per_launch"] if dom else None},
           "per_step_total_bytes": int(total), "kernels": [k for k in kernels if k["avg_us_under_pmc"] * k["launches"] / steps > 5.0]}
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", tag + "_pmc_traffic.json")
    json.dump(doc, open(out, "w"), indent=1)
    print(out, "dominant:", doc["dominant"], "step total GB: %.2f" % (total / 1e9))


if __name__ == "__main__":
    main()
