"""Turn the rocprofv3 --kernel-trace --stats databases written by tools/profile_bench.sh into the committed summaries:

  profiles/<tag>_bench_bf16_kernel_stats.csv        inference bench (the headline command)
  profiles/<tag>_train_bf16_kernel_stats.csv        bench.py --mode train
  profiles/<tag>_train_video_bf16_kernel_stats.csv  bench.py --mode train_video
  profiles/<tag>_*_under_rocprof.log                the bench line printed under the profiler
  profiles/<tag>_bench_bf16.json                    the bench line of a plain run on the same box

  python tools/stats_summary.py gpurun_out/r2 r2
"""
import csv
import glob
import os
import shutil
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(names):
    """rocprofv3 leaves names with bf16 / fp16 template arguments mangled (its demangler predates DF16b / DF16_): rewrite those
    two as vendor types and let c++filt do the rest"""
    import subprocess
    names = [n[:-3] if n.endswith(".kd") else n for n in names]
    fixed = [n.replace("DF16b", "u14__hip_bfloat16").replace("DF16_", "u8_Float16") for n in names]
    try:
        out = subprocess.run(["c++filt"], input="\n".join(fixed), capture_output=True, text=True, check=True).stdout.splitlines()
        return out if len(out) == len(names) else names
    except (OSError, subprocess.CalledProcessError):
        return names


def dump(db_dir, out_csv):
    files = glob.glob(os.path.join(db_dir, "**", "*_results.db"), recursive=True)
    if not files:
        print("no database under", db_dir)
        return
    c = sqlite3.connect(files[0])
    # the symbol table's kernel_name is always the mangled name; display_name is sometimes a wrong demangling (see demangle)
    rows = c.execute("select S.kernel_name, count(*), sum(K.end - K.start) / 1000.0, sum(K.end - K.start) / count(*) / 1000.0, "
                     "sum(K.end - K.start) * 100.0 / (select sum(A.end - A.start) from rocpd_kernel_dispatch A) "
                     "from rocpd_kernel_dispatch K inner join rocpd_info_kernel_symbol S on S.id = K.kernel_id and S.guid = K.guid "
                     "group by S.kernel_name order by 3 desc").fetchall()
    rows = [(n,) + tuple(r[1:]) for n, r in zip(demangle([r[0] for r in rows]), rows)]
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
        for name, calls, total, avg, pct in rows:
            w.writerow([name, int(calls), round(total, 3), round(avg, 3), round(pct, 4)])
    print(out_csv, len(rows), "kernels; top:", rows[0][0][:80], "avg %.1f us" % rows[0][3])


def last_json_line(path):
    if not os.path.exists(path):
        return None
    for line in reversed(open(path).read().splitlines()):
        if line.startswith("{"):
            return line
    return None


def main():
    base, tag = sys.argv[1], sys.argv[2]
    P = os.path.join(ROOT, "profiles")
    for sub, name, log in [("stats", "bench", "bench_stats.log"), ("train", "train_replay", "train_stats.log"),
                           ("train_eager", "train_eager", "train_eager_stats.log"),
                           ("train_video", "train_video", "train_video_stats.log")]:
        dump(os.path.join(base, sub), os.path.join(P, "%s_%s_bf16_kernel_stats.csv" % (tag, name)))
        line = last_json_line(os.path.join(base, log))
        if line:
            open(os.path.join(P, "%s_%s_bf16_under_rocprof.log" % (tag, name)), "w").write(line + "\n")
    line = last_json_line(os.path.join(base, "bench_plain.log"))
    if line:
        open(os.path.join(P, "%s_bench_bf16.json" % tag), "w").write(line + "\n")
    for f in glob.glob(os.path.join(P, tag + "_*")):
        shutil.copy(f, base)
    for d in ("stats", "train", "train_eager", "train_video", "pmc_fetch", "pmc_write"):      # the databases are tens of MiB each: scratch only
        shutil.rmtree(os.path.join(base, d), ignore_errors=True)


if __name__ == "__main__":
    main()
