"""Host time of one lib.call (Python -> ctypes -> hipLaunchKernel) with a tiny kernel and a many-argument one; the batch-1 forward is 143 launches."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtfs_net_amd import lib

x, y = torch.zeros(256, device="cuda"), torch.zeros(256, device="cuda")
for name, args in (("rtfs_axpy", (x, 1.0, y, 256)),):
    for _ in range(200):
        lib.call(name, *args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5000):
        lib.call(name, *args)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name}: {1e6 * (t1 - t0) / 5000:.2f} us of host time per call (queue drained {1e3 * (t2 - t1):.1f} ms later)")
