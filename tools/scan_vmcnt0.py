"""Static scan of `hipcc -S` output: per kernel, the loops (label ... backward branch to it) that contain global / buffer loads together with
`s_waitcnt vmcnt(0)` (a full drain of every outstanding load and store inside a loop that is supposed to keep loads in flight) or scratch accesses.
usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -S --cuda-device-only x.hip -o x.s; python tools/scan_vmcnt0.py x.s [name filter]"""
import re
import subprocess
import sys

lines = open(sys.argv[1]).read().split("\n")
filt = sys.argv[2] if len(sys.argv) > 2 else ""
kern, start = None, 0
kernels = []
for i, l in enumerate(lines):
    m = re.match(r"^(_Z\w+):", l)
    if m:
        kern, start = m.group(1), i
    if l.startswith(".Lfunc_end") and kern:
        kernels.append((kern, start, i))
        kern = None
for name, a, b in kernels:
    body = lines[a:b]
    labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\w+):", l))}
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\w+)|s_branch\s+(\.LBB\w+)", l)
        if m:
            t = m.group(1) or m.group(2)
            if t in labels and labels[t] < i:
                loops.append((labels[t], i))
    out = []
    for la, lb in loops:
        seg = body[la:lb]
        loads = sum(1 for l in seg if re.search(r"\b(global_load|buffer_load)", l))
        stores = sum(1 for l in seg if re.search(r"\b(global_store|buffer_store)", l))
        drains = sum(1 for l in seg if "vmcnt(0)" in l)
        scr = sum(1 for l in seg if "scratch_" in l)
        mfma = sum(1 for l in seg if "v_mfma" in l)
        if loads and (drains or scr):
            out.append(f"    loop of {lb - la} lines: {loads} loads, {stores} stores, {mfma} MFMAs, {drains} x vmcnt(0), {scr} scratch accesses")
    if out:
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()[:150]
        if filt in dem:
            print(dem)
            print("\n".join(out))
