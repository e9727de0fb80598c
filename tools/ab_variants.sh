#!/bin/bash
# Same-box A/B of compile-time variants of the 3x3 kernels: tools/ab_variants.sh name "flags" [name "flags" ...]
#   -> tools/_ab/lib<name>.so = the current objects (build/*.o, python -c "import __graft_entry__ as g; g.build()" first) with
#      conv3x3_pc.hip and conv3x3_pipe.hip recompiled under the given -D flags.  Time them on the GPU box with
#      python tools/layer_times.py tools/_ab/lib<name>.so
#   FILES="upconv2x2" tools/ab_variants.sh ... recompiles those sources instead (default: conv3x3_pc conv3x3_pipe)
set -e
FILES=${FILES:-"conv3x3_pc conv3x3_pipe"}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/tools/_ab" "$ROOT/build/ab"
pids=()
names=()
while [ $# -ge 2 ]; do
  NAME=$1; FLAGS=$2; shift 2
  names+=("$NAME")
  for f in $FILES; do
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 $FLAGS -c "$ROOT/uncltmo_amd/csrc/$f.hip" -o "$ROOT/build/ab/$f.$NAME.o" 2>/dev/null &
    pids+=($!)
  done
done
for p in "${pids[@]}"; do wait $p; done
for NAME in "${names[@]}"; do
  OBJS=$(ls "$ROOT"/build/*.hip.o)
  VOBJS=""
  for f in $FILES; do
    OBJS=$(echo "$OBJS" | grep -v "/$f.hip.o")
    VOBJS="$VOBJS $ROOT/build/ab/$f.$NAME.o"
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $VOBJS -o "$ROOT/tools/_ab/lib$NAME.so"
  echo built tools/_ab/lib$NAME.so
done
