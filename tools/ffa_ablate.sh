#!/bin/bash
# usage: tools/ffa_ablate.sh [ws6|lb] <mask> ...   -> exp/ffa<mask>/librtfs_hip.so: dualpath.hip rebuilt with -DFFA_ABL=<mask>, linked with the objects of the current build
# ABL_FLAGS: extra hipcc flags (e.g. -DUW_TIMING for tools/uw_clock.py)
# (timing-only builds of unfold_ffa_kernel: 1 no staging, 2 no write-back, 4 no fetch, 8 no barriers; combine by adding)
cd "$(dirname "$0")/.."
MACRO=FFA_ABL; PFX=ffa; SRC=dualpath
if [ "$1" = ws6 ]; then MACRO=WS6_ABL; PFX=ws6; shift; fi  # unfold_ws6_kernel: 1 no staging, 2 no write-back, 4 no fetch, 8 no barriers, 16 no fragment reads
if [ "$1" = lb ]; then MACRO=LB_ABL; PFX=lb; SRC=bwd_seq; shift; fi  # sru_layer_bwd_kernel (bwd_seq.hip): 1 no dW MFMAs, 2 no dX MFMAs, 4 no recurrence, 8 no dX stores, 16 no loads
for m in "$@"; do
  out=exp/$PFX$m; mkdir -p $out
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -D$MACRO=$m $ABL_FLAGS -c rtfs_net_amd/csrc/$SRC.hip -o $out/$SRC.o &
done
wait
for m in "$@"; do
  out=exp/$PFX$m
  objs=$(ls rtfs_net_amd/csrc/*.o | grep -v $SRC.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $out/$SRC.o -o $out/librtfs_hip.so && rm -f $out/$SRC.o
done
ls -la exp/$PFX*/librtfs_hip.so
