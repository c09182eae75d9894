#!/bin/bash
# usage: tools/build_variant.sh <name> [extra hipcc flags...]   -> exp/<name>/librtfs_hip.so  (same-box A/B builds; select with RTFS_HIP_LIB)
name=$1; shift
out=exp/$name; mkdir -p $out
for f in rtfs_net_amd/csrc/*.hip; do
  o=$out/$(basename ${f%.hip}).o
  extra=""; case $(basename $f) in vp.hip|tfar.hip|vp_train.hip|vp_attn.hip) extra="-fno-slp-vectorize";; esac  # (rtfs_net_amd/build.py EXTRA_FLAGS)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast $extra "$@" -c $f -o $o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $out/*.o -o $out/librtfs_hip.so && rm -f $out/*.o && ls -la $out/librtfs_hip.so
