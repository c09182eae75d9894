#!/bin/bash
# usage: tools/mkvar.sh <name> <file-stem (gemm, dualpath, ...)> <sed-expr>   -> exp/<name>/librtfs_hip.so with csrc/<stem>.hip modified by the
# sed expression (throw-away same-box A/B builds; select with RTFS_HIP_LIB).  Needs the regular build's objects (rtfs_net_amd/csrc/*.o).
name=$1; stem=$2; expr=$3
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/exp/$name; mkdir -p $out
sed -e "$expr" $root/rtfs_net_amd/csrc/$stem.hip > $root/rtfs_net_amd/csrc/${stem}_var_$name.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -c $root/rtfs_net_amd/csrc/${stem}_var_$name.hip -o $out/$stem.o || exit 1
rm $root/rtfs_net_amd/csrc/${stem}_var_$name.hip
objs=$(ls $root/rtfs_net_amd/csrc/*.o | grep -v "/$stem.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $out/$stem.o $objs -o $out/librtfs_hip.so && rm $out/$stem.o && echo built $out
