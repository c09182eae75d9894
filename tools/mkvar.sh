#!/bin/bash
# usage: mkvar.sh <name> <sed-expr>   -> exp/<name>/librtfs_hip.so with gemm.hip modified by the sed expression
name=$1; expr=$2
out=/root/repo/exp/$name; mkdir -p $out
sed -e "$expr" /root/repo/rtfs_net_amd/csrc/gemm.hip > /root/repo/rtfs_net_amd/csrc/gemm_var_$name.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -c /root/repo/rtfs_net_amd/csrc/gemm_var_$name.hip -o $out/gemm.o || exit 1
rm /root/repo/rtfs_net_amd/csrc/gemm_var_$name.hip
objs=$(ls /root/repo/rtfs_net_amd/csrc/*.o | grep -v "/gemm.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $out/gemm.o $objs -o $out/librtfs_hip.so && rm $out/gemm.o && echo built $out
