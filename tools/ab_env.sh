#!/bin/bash
# run on the GPU box: per-layer times of the in-tree library under two settings of one environment switch, interleaved
# tools/ab_env.sh <tag> <VAR> "<values>" <reps>
TAG=$1; VAR=$2; VALS=$3; REPS=${4:-3}
mkdir -p gpurun_out/$TAG
for rep in $(seq 1 $REPS); do
  for v in $VALS; do
    env $VAR=$v python tools/layer_times.py > gpurun_out/$TAG/ab_${VAR}${v}_$rep.log 2>&1
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
print("%-34s" % "layer (min over reps, ms)" + "".join("%18s" % n for n in names))
tot = collections.defaultdict(float)
for layer, r in rows.items():
    if max(min(v) for v in r.values()) < 0.01: continue
    print("%-34s" % layer + "".join("%18.3f" % min(r[n]) for n in names))
    for n in names: tot[n] += min(r[n])
print("%-34s" % "sum" + "".join("%18.3f" % tot[n] for n in names))
PY
