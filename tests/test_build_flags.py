"""CPU: the side-stream kernels must not contain packed-fp32 instructions that swap operand halves (`v_pk_*_f32 ... op_sel:[..]`): next to the
main stream's bf16 MFMA kernels those returned wrong low halves (rule 10).  rtfs_net_amd/build.py builds the video-branch
sources without the SLP vectoriser for that reason; this test compiles them to assembly with the build's own flags and looks."""
import os
import re
import shutil
import subprocess

import pytest

from rtfs_net_amd import build

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("src,kernels", [("vp.hip", None), ("vp_train.hip", None), ("vp_attn.hip", None), ("tfar.hip", ["caf_video_kernel", "caf_video_bwd_kernel"])])
def test_side_stream_kernels_have_no_packed_op_sel(tmp_path, src, kernels):
    """kernels = None: EVERY kernel of the file (vp.hip / vp_train.hip / vp_attn.hip hold nothing but the video branch, which the inference path and
    the training step both run on the side stream); tfar.hip: its two CAF video-side kernels (the depth-wise kernels of that file run on the main
    stream and use native 4-vectors)"""
    assert "-fno-slp-vectorize" in build.EXTRA_FLAGS[src]
    out = tmp_path / (src + ".s")
    subprocess.run([HIPCC, *build.FLAGS, *build.EXTRA_FLAGS[src], "-S", "--cuda-device-only", os.path.join(build.CSRC, src), "-o", str(out)], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    kern, bad, seen = None, {}, set()
    for line in out.read_text().split("\n"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kern = m.group(1)
        if kern and (kernels is None or any(k in kern for k in kernels)):
            seen.add(kern)
            if re.search(r"v_pk_\w+_f32", line) and re.search(r"op_sel:\[", line):
                bad[kern] = bad.get(kern, 0) + 1
    assert seen, "kernels not found in the assembly"
    assert not bad, bad
