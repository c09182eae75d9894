"""Where the torch glue of a training step sits on the device timeline: runs of consecutive non-rtfs kernels (by start time, all streams) of the last step of a
`rocprofv3 --kernel-trace --output-format csv` trace, with their span and the rtfs kernels on either side.
    python tools/train_glue_clusters.py /tmp/tg"""
import csv, glob, sys

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path)))
marks = [i for i, r in enumerate(rows) if "adamw_clip_kernel" in r[2]]
seg = rows[marks[-2] + 1:marks[-1] + 1]
clusters, cur, prev = [], [], "(step start)"
for s, e, n in seg:
    if "rtfs::" in n:
        if cur:
            clusters.append((cur, prev, n))
            cur = []
        prev = n
    else:
        cur.append((s, e, n))
if cur:
    clusters.append((cur, prev, "(step end)"))
tot = 0
out = []
for ks, before, after in clusters:
    span = ks[-1][1] - ks[0][0]
    busy = sum(e - s for s, e, _ in ks)
    tot += busy
    out.append((span, len(ks), busy, before, after))
print(f"{len(clusters)} runs of torch kernels, {sum(len(c[0]) for c in clusters)} launches, {1e-6 * tot:.2f} ms of kernel time")
out.sort(reverse=True)
for span, n, busy, before, after in out[:16]:
    print(f"{1e-3 * span:8.1f} us span {n:3d} launches {1e-3 * busy:7.1f} us busy   after {before[:48]:48s} before {after[:48]}")
