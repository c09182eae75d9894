"""rtfs_net_amd: MI355X-native (gfx950) implementation of RTFS-Net's separation forward path.

`models`  host-side mirror of the reference's `src.models` API (AVNet / RTFSNet, get, register_model)
`csrc`    hand-written HIP kernels + the C-ABI of include/rtfs_hip.h (built into librtfs_hip.so)
`lib`     ctypes binding of that C-ABI (fails loudly when the library is missing)
"""
from .models import AVNet, RTFSNet, get, register_model  # noqa: F401

__all__ = ["AVNet", "RTFSNet", "get", "register_model"]
