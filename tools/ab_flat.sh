#!/bin/bash
# run on the GPU box: per-layer times with flat tiles off / on (same library, env switch), three interleaved repetitions
TAG=${1:-abflat}
mkdir -p gpurun_out/$TAG
for rep in 1 2 3; do
  for f in 0 1; do
    UNCL_FLAT=$f python tools/layer_times.py > gpurun_out/$TAG/ab_flat${f}_$rep.log 2>&1
  done
done
python3 - <<PY
import glob, re, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/$TAG/ab_*_*.log")):
    name = re.match(r".*/ab_(.*)_\d+\.log", f).group(1)
    for l in open(f):
        m = re.match(r"\s*(\d+) (\S+)\s+([\d.]+) ms", l)
        if m: rows[m.group(2)][name].append(float(m.group(3)))
names = sorted({n for r in rows.values() for n in r})
print("%-34s" % "layer (min of 3, ms)" + "".join("%8s" % n for n in names))
tot = collections.defaultdict(float)
for layer, r in rows.items():
    if max(min(v) for v in r.values()) < 0.01: continue
    print("%-34s" % layer + "".join("%8.3f" % min(r[n]) for n in names))
    for n in names: tot[n] += min(r[n])
print("%-34s" % "sum" + "".join("%8.3f" % tot[n] for n in names))
PY
