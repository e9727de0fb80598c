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

# up_path.3.conv.conv (loader-fused up-conv): the producer/consumer kernel, MODE 4; the four-wave kernel when UNCL_PC=0
DOMINANT = ("conv3x3_pc_kernel<__hip_bfloat16, 1, 4, 4,", "conv3x3_pipe_kernel<__hip_bfloat16, 1, 4, 4, 4,", "conv3x3_pipe_kernel<1, 4, 4, 4")


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
        # counters_collection with the symbol table's (always mangled) kernel_name instead of display_name
        for r in db.execute("select S.kernel_name, K.grid_size_x * K.grid_size_y * K.grid_size_z, sum(E.value), K.start, K.end "
                            "from rocpd_pmc_event E inner join rocpd_info_pmc I on I.id = E.pmc_id and I.guid = E.guid "
                            "inner join rocpd_kernel_dispatch K on K.event_id = E.event_id and K.guid = E.guid "
                            "inner join rocpd_info_kernel_symbol S on S.id = K.kernel_id and S.guid = K.guid "
                            "where I.name = ? group by K.dispatch_id order by K.dispatch_id", (counter,)):
            add(*r)
    return rows


def _demangled(rows):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from stats_summary import demangle
    keys = list(rows.keys())
    names = demangle([k[0] for k in keys])
    return collections.OrderedDict(((n, k[1]), rows[k]) for n, k in zip(names, keys))


def main():
    base, tag = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5        # timed + warm-up steps of the PMC run
    fetch = _demangled(read(os.path.join(base, "pmc_fetch"), "FETCH_SIZE"))
    write = _demangled(read(os.path.join(base, "pmc_write"), "WRITE_SIZE"))
    kernels, total = [], 0.0
    for key, (fs, n, us) in fetch.items():
        ws = write.get(key, [0.0, 1, 0.0])
        f_kib, w_kib = fs / n, ws[0] / max(ws[1], 1)
        per_launch = (2.0 * f_kib + w_kib) * 1024.0
        total += per_launch * n / steps
        kernels.append({"kernel": key[0], "grid_x": key[1], "launches": n, "avg_us_under_pmc": round(us / n, 1),
                        "FETCH_SIZE_KiB": round(f_kib, 1), "WRITE_SIZE_KiB": round(w_kib, 1), "hbm_bytes_per_launch": int(per_launch)})
    kernels.sort(key=lambda k: -k["avg_us_under_pmc"] * k["launches"])
    dom = next((k for k in kernels if any(d in k["kernel"] for d in DOMINANT)), None)
    doc = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separately, WRITE_SIZE) -- python3 bench.py --steps 4 "
                      "--warmup 1 --no-cpu --no-train --no-exclusive (tools/profile_bench.sh, tools/pmc_summary.py)",
           "correction": "gfx950: FETCH_SIZE counts wide coalesced reads at 1/2 (MI355X_MICROARCH.md, HBM section) -> hbm_bytes = "
                         "(2*FETCH_SIZE + WRITE_SIZE) * 1024; counters are KiB per launch, averaged over the run's launches",
           "dominant": {"kernel": (dom["kernel"] if dom else DOMINANT[0]) + " @ up_path.3.conv.conv", "grid_x": dom["grid_x"] if dom else None,
                        "hbm_bytes_per_launch": dom["hbm_bytes_per_launch"] if dom else None},
           "per_step_total_bytes": int(total), "kernels": [k for k in kernels if k["avg_us_under_pmc"] * k["launches"] / steps > 5.0]}
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", tag + "_pmc_traffic.json")
    json.dump(doc, open(out, "w"), indent=1)
    print(out, "dominant:", doc["dominant"], "step total GB: %.2f" % (total / 1e9))


if __name__ == "__main__":
    main()
