"""Scan gfx950 assembly (hipcc -S --cuda-device-only) for points where more LDS / scalar-memory operations are in flight than the 4-bit
lgkmcnt counter can represent (15), or more vector-memory loads than the 6-bit vmcnt (63): straight-line count per kernel, reset by the
matching s_waitcnt.  usage: python tools/scan_waitcnt.py file.s [...]"""
import re
import sys

LGKM = re.compile(r"^\s*(ds_|s_load|s_buffer_load|s_sendmsg|buffer_.*\slds|global_load_lds)")
VM = re.compile(r"^\s*(global_load|buffer_load|flat_load|global_atomic.*\sglc|scratch_load)")
WAIT = re.compile(r"^\s*s_waitcnt\s+(.*)")
for path in sys.argv[1:]:
    kernel, lg, vm, worst = None, 0, 0, {}
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kernel, lg, vm = m.group(1), 0, 0
            continue
        if kernel is None:
            continue
        if re.match(r"^\.LBB", line):
            pass  # loop back-edges re-enter with whatever is pending: keep the straight-line count
        w = WAIT.match(line)
        if w:
            a = re.search(r"lgkmcnt\((\d+)\)", w.group(1))
            b = re.search(r"vmcnt\((\d+)\)", w.group(1))
            if a:
                lg = min(lg, int(a.group(1)))
            if b:
                vm = min(vm, int(b.group(1)))
            continue
        if "s_barrier" in line and False:
            continue
        if LGKM.match(line):
            lg += 1
            if lg > 15:
                worst[kernel] = max(worst.get(kernel, (0, 0))[0], lg), worst.get(kernel, (0, 0))[1]
        elif VM.match(line):
            vm += 1
            if vm > 63:
                worst[kernel] = worst.get(kernel, (0, 0))[0], max(worst.get(kernel, (0, 0))[1], vm)
    for k, (a, b) in sorted(worst.items()):
        print(f"{path.split('/')[-1]:14s} lgkm {a:3d} vm {b:3d}  {k[:110]}")
