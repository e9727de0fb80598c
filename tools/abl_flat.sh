#!/bin/bash
# run on the GPU box: per-layer times under every tools/_ab/libabl*.so (compile-time ablations of conv3x3_flat.hip, built with
#   FILES=conv3x3_flat tools/ab_variants.sh abl0 "-DUNCL_FL_ABL_MASK=0" abl16 "-DUNCL_FL_ABL_MASK=16" ...)
TAG=${1:-ablflat}
mkdir -p gpurun_out/$TAG
for lib in tools/_ab/libabl*.so; do
  n=$(basename $lib .so); n=${n#libabl}
  UNCL_FLAT=${UNCL_FLAT:-1} python tools/layer_times.py $lib > gpurun_out/$TAG/abl_$n.log 2>&1
done
python3 - <<PY
import glob, re, collections
rows = collections.defaultdict(dict)
names = []
for f in sorted(glob.glob("gpurun_out/$TAG/abl_*.log"), key=lambda x: int(re.findall(r"abl_(\d+)", x)[0])):
    name = re.findall(r"abl_(\d+)", f)[0]
    names.append(name)
    for l in open(f):
        m = re.match(r"\s*(\d+) (\S+)\s+([\d.]+) ms", l)
        if m: rows[m.group(2)][name] = float(m.group(3))
print("%-34s" % "layer (ms)" + "".join("%8s" % n for n in names))
for layer, r in rows.items():
    if max(r.values()) < 0.01: continue
    print("%-34s" % layer + "".join("%8.3f" % r.get(n, 0) for n in names))
PY
