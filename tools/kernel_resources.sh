#!/bin/bash
# usage: tools/kernel_resources.sh <file.hip> [name filter]   -> VGPRs / AGPRs / spills / LDS / occupancy per kernel (hipcc -Rpass-analysis=kernel-resource-usage)
src=$1; filt=${2:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Rpass-analysis=kernel-resource-usage $EXTRA -c "$src" -o /dev/null 2>&1 |
  awk '/Function Name:/ {name=$(NF-1)} / VGPRs:/ {v=$(NF-1)} / AGPRs:/ {a=$(NF-1)} /VGPRs Spill:/ {s=$(NF-1)} /Occupancy/ {o=$(NF-1)} /LDS Size/ {l=$(NF-1); print name, "vgpr", v, "agpr", a, "spill", s, "lds", l, "occ", o}' |
  c++filt | cut -c1-400 | grep -E "$filt"
