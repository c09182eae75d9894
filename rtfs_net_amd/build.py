"""Build librtfs_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m rtfs_net_amd.build [--force]
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["gemm.hip", "dualpath.hip", "attention.hip", "tfar.hip", "stft.hip", "bwd_elem.hip", "bwd_gemm.hip", "bwd_seq.hip", "bwd_attn.hip",
           "bwd_misc.hip", "spread.hip", "loss.hip", "vp.hip", "vp_train.hip", "vp_attn.hip", "lip.hip", "optim.hip", "views.hip", "bwd_dw.hip"]
LIB = os.path.join(HERE, "librtfs_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast"]
# The video-branch kernels run on a SIDE stream, i.e. next to the main stream's bf16 MFMA kernels on the same CUs in the bf16 / split-bf16
# modes.  Packed-fp32 instructions that swap operand halves (v_pk_*_f32 ... op_sel), which hipcc's SLP vectoriser produces by the hundred in
# these kernels, returned wrong low halves next to bf16 MFMA traffic (rule 10: first seen inside attn_qkv_kernel): with the
# plain-bf16 forward 20-29 of 30 runs differed from the first by up to 2e-3, 0 of 30 with the video branch on the main stream, 0 of 30
# with these files built without the SLP vectoriser (vp_block_kernel: 256 -> 0 such instructions, caf_video_kernel: 55 -> 0).  The depth-wise
# kernels of tfar.hip use native 4-vectors for their packed FMAs and are not affected by the switch.
EXTRA_FLAGS = {src: ["-fno-slp-vectorize"] for src in ("vp.hip", "tfar.hip", "vp_train.hip", "vp_attn.hip")}


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.abspath(__file__)]  # (the flags live in this file)
    return any(os.path.getmtime(d) > t for d in deps)


def _obj_stale(src_path: str, obj: str) -> bool:
    """an object is rebuilt when its source, any header of csrc/ or include/, or this file (the flags) is newer"""
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    inc = os.path.join(os.path.dirname(HERE), "include")
    deps = [src_path, os.path.abspath(__file__)] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    if os.path.isdir(inc):
        deps += [os.path.join(inc, f) for f in os.listdir(inc)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(obj)
        if not force and not _obj_stale(os.path.join(CSRC, src), obj):
            continue
        cmd = [hipcc, *FLAGS, *EXTRA_FLAGS.get(src, []), "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
