#!/bin/bash
# Build the library of another commit next to the current one for same-box A/B timing:
#   tools/ab_build.sh <commit> <name>   ->  tools/_ab/lib<name>.so   (csrc of that commit, current headers must match)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$1; NAME=$2
TMP=$(mktemp -d)
git -C "$ROOT" archive "$C" uncltmo_amd/csrc include | tar -x -C "$TMP"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -shared -I"$TMP/include" -I"$TMP/uncltmo_amd/csrc" -o "$ROOT/tools/_ab/lib$NAME.so" "$TMP"/uncltmo_amd/csrc/*.hip
rm -rf "$TMP"
echo built tools/_ab/lib$NAME.so
