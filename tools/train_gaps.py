"""GPU idle time inside the training step: `rocprofv3 --kernel-trace --output-format csv` of `bench.py --mode train`, then the union of all kernel
intervals of the LAST step (both streams) - busy time, idle time, and the largest gaps with the kernels on either side.
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tg -o t -- python /root/repo/bench.py --mode train --steps 4 --warmup 2 --no-cpu-baseline
    python tools/train_gaps.py /tmp/tg [marker kernel, default adamw_clip_kernel; the inference forward: istft_ola_kernel]"""
import csv, glob, sys

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]))
rows.sort()
# steps are delimited by the optimizer kernel (adamw_clip_kernel) - take the interval between the last two
marker = sys.argv[2] if len(sys.argv) > 2 else "adamw_clip_kernel"  # a kernel that runs once per step (inference: istft_ola_kernel)
marks = [i for i, r in enumerate(rows) if marker in r[2]]
if len(marks) < 2:
    sys.exit("no step markers")
a, b = marks[-2] + 1, marks[-1] + 1
seg = rows[a:b]
t0, t1 = seg[0][0], max(r[1] for r in seg)
busy, cur_end, gaps = 0, seg[0][0], []
last = seg[0]
for s, e, n in seg:
    if s > cur_end:
        gaps.append((s - cur_end, last[2], n))
        busy += 0
        cur_start = s
    if e > cur_end:
        busy += e - max(s, cur_end)
        cur_end = e
        last = (s, e, n)
print(f"step span {1e-6 * (t1 - t0):.2f} ms, {len(seg)} launches, busy {1e-6 * busy:.2f} ms, idle {1e-6 * (t1 - t0 - busy):.2f} ms in {len(gaps)} gaps")
import collections
hist = collections.Counter()
for g, _, _ in gaps:
    hist["<2us" if g < 2000 else "<5us" if g < 5000 else "<20us" if g < 20000 else ">=20us"] += g
print({k: f"{1e-6 * v:.2f} ms" for k, v in hist.items()})
gaps.sort(reverse=True)
for g, before, after in gaps[:25]:
    print(f"{1e-3 * g:8.1f} us  after {before[:60]:60s} before {after[:60]}")
