#!/usr/bin/env python
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as gpurun requires).

FETCH_SIZE / WRITE_SIZE are reported in KiB.  MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reads exactly 1/2
of the bytes of a wide coalesced streaming read (16 B/lane) -> reads are DOUBLED here; WRITE_SIZE is used as reported
(uncalibrated per the guide).  Output: average bytes per launch per kernel.
"""
import collections
import csv
import sys


def per_kernel(path, counter):
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            tot[r["Kernel_Name"]] += float(r["Counter_Value"])
            cnt[r["Kernel_Name"]] += 1
    return {k: tot[k] / cnt[k] for k in tot}, cnt


def main(fetch_csv, write_csv, out=None):
    f, n = per_kernel(fetch_csv, "FETCH_SIZE")
    w, _ = per_kernel(write_csv, "WRITE_SIZE")
    rows = []
    for k in f:
        rd = 2.0 * f[k] * 1024.0
        wr = w.get(k, 0.0) * 1024.0
        rows.append((rd + wr, k, n[k], rd, wr))
    rows.sort(reverse=True)
    lines = ["# avg HBM bytes per launch: read = 2 x FETCH_SIZE KiB (gfx950 correction), write = WRITE_SIZE KiB",
             f"{'kernel':100s} {'launches':>8s} {'read_MB':>10s} {'write_MB':>10s} {'total_MB':>10s}"]
    for tot, k, c, rd, wr in rows[:40]:
        lines.append(f"{k[:100]:100s} {c:8d} {rd / 1e6:10.1f} {wr / 1e6:10.1f} {tot / 1e6:10.1f}")
    passes = [c for _, k, c, _, _ in rows if "stft_kernel" in k and "istft" not in k]  # rtfs::stft_kernel runs once per forward / step
    total = sum(tot * c for tot, _, c, _, _ in rows)
    lines.append(f"# all {len(rows)} kernels, launches x bytes: {total / 1e9:.1f} GB over {passes[0] if passes else '?'} passes"
                 + (f" = {total / 1e9 / passes[0]:.1f} GB per forward / step" if passes else ""))
    text = "\n".join(lines) + "\n"
    (open(out, "w") if out else sys.stdout).write(text)
    if out:
        import json
        import os

        js = {k: {"bytes_per_launch": tot, "read_bytes": rd, "write_bytes": wr, "launches": c} for tot, k, c, rd, wr in rows}
        # what these bytes describe: the kernel sources of the run.  bench.py reports `traffic` only while the roofline kernel's source file still has this hash
        import hashlib

        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        js["_source"] = {rel: hashlib.sha256(open(os.path.join(root, rel), "rb").read()).hexdigest()
                         for rel in ("rtfs_net_amd/csrc/dualpath.hip", "rtfs_net_amd/csrc/common.h")}
        json.dump(js, open(os.path.splitext(out)[0] + ".json", "w"), indent=0)  # every kernel; tools/pmc_hbm.sh copies the forward one to pmc_traffic.json


if __name__ == "__main__":
    main(*sys.argv[1:4])
