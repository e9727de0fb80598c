#!/bin/bash
# register / scratch summary of every kernel of one source: tools/kres.sh conv3x3_flat [extra flags]
f=$1; shift
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 "$@" -c uncltmo_amd/csrc/$f.hip -o /tmp/kres_$f.o -Rpass-analysis=kernel-resource-usage 2>&1 |
  awk '/error/ {print} /Function Name/ {n=$NF} / VGPRs:/ {v=$(NF-1)} /ScratchSize/ {s=$(NF-1)} /VGPRs Spill/ {sp=$(NF-1)} /Occupancy/ {o=$(NF-1)} /LDS Size/ {printf "%-110s vgpr %3s scratch %4s spill %3s occ %s\n", n, v, s, sp, o}' | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//' | while read l; do echo "$l" | c++filt 2>/dev/null || echo "$l"; done
