"""Summarise the SQ counter pass of tools/profile_bench.sh into profiles/<tag>_pmc_sq.json: per kernel of the bench step the
vector / matrix instruction counts, their ratio (the matrix and the vector pipe share a SIMD's issue port: MI355X_MICROARCH.md,
cycle constants), the cycles the matrix pipe was busy and the wave / busy cycles, per launch.

  python tools/pmc_sq_summary.py gpurun_out/r3a r3a [steps_in_pmc_run]
"""
import collections
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import _demangled, read  # noqa: E402

COUNTERS = ["SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES"]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    base, tag = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    per = collections.OrderedDict()
    for c in COUNTERS:
        for key, (val, n, us) in _demangled(read(os.path.join(base, "pmc_sq"), c)).items():
            e = per.setdefault(key, {"launches": n, "us": us})
            e[c] = val
    # clock pass (its own run): GRBM_GUI_ACTIVE is summed over the 8 XCDs -> effective clock = count / 8 / launch time; reads
    # high on launches shorter than ~0.3 ms (MI355X_MICROARCH.md, DVFS give-back)
    if os.path.isdir(os.path.join(base, "pmc_clk")):
        for key, (val, n, us) in _demangled(read(os.path.join(base, "pmc_clk"), "GRBM_GUI_ACTIVE")).items():
            if key in per and us > 0:
                per[key]["clk_ghz"] = val / 8.0 / (us * 1000.0)
                per[key]["clk_us"] = us / max(n, 1)
    kernels = []
    for (name, grid), e in per.items():
        n = max(e["launches"], 1)
        k = {"kernel": name, "grid": grid, "launches_per_step": round(e["launches"] / steps, 2), "avg_us": round(e["us"] / n, 2)}
        for c in COUNTERS:
            if c in e:
                k[c + "_per_launch"] = round(e[c] / n, 1)
        if "clk_ghz" in e:
            k["effective_clock_ghz"] = round(e["clk_ghz"], 3)
            k["avg_us_in_clock_pass"] = round(e["clk_us"], 2)
        if e.get("SQ_INSTS_MFMA"):
            k["valu_per_mfma"] = round(e.get("SQ_INSTS_VALU", 0.0) / e["SQ_INSTS_MFMA"], 3)
            if e.get("SQ_BUSY_CYCLES"):
                # SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per 32x32x16 bf16 MFMA, summed over SIMDs); SQ_BUSY_CYCLES is per
                # shader engine / XCD aggregate: report the raw ratio, not a utilisation claim
                k["mfma_busy_over_sq_busy"] = round(e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / e["SQ_BUSY_CYCLES"], 4)
        kernels.append(k)
    kernels.sort(key=lambda k: -k["avg_us"] * k["launches_per_step"])
    out = {"source": "rocprofv3 --kernel-trace --pmc " + " ".join(COUNTERS) + " -- python3 bench.py --steps 3 --warmup 1 --no-cpu "
                     "--no-train --no-exclusive --no-layers (tools/profile_bench.sh)", "steps_in_run": steps, "kernels": kernels}
    path = os.path.join(ROOT, "profiles", "%s_pmc_sq.json" % tag)
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path, len(kernels), "kernels")
    for k in kernels[:10]:
        print("  %-90s %8.1f us  VALU/MFMA %s  clock %s GHz" % (k["kernel"][:90], k["avg_us"], k.get("valu_per_mfma"),
                                                                k.get("effective_clock_ghz")))


if __name__ == "__main__":
    main()
