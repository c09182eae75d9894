#!/usr/bin/env python
"""MFMA utilisation per kernel from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE, SQ_WAVE_CYCLES, SQ_BUSY_CYCLES):
busy % = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs (both counters are sums
over their instances).  SQ_VALU_MFMA_BUSY_CYCLES counts 64 per v_mfma_f32_32x32x2_f32 and 32 per v_mfma_f32_16x16x4_f32 (checked
against the instruction counts of the layer-0 GEMM), i.e. it is the time the matrix pipe needs at its peak rate.
usage: pmc_mfma_util.py <counter_collection.csv> [out.txt]"""
import collections
import csv
import sys

tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[k][r["Counter_Name"]] += 1
rows = []
for k, c in tot.items():
    n = cnt[k]["GRBM_GUI_ACTIVE"]
    if not n or "SQ_VALU_MFMA_BUSY_CYCLES" not in c:
        continue
    cyc = c["GRBM_GUI_ACTIVE"] / n / 8.0
    busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / n / 1024.0
    if busy <= 0:
        continue
    rows.append((busy * n, k, n, cyc, busy, 100.0 * busy / cyc))
rows.sort(reverse=True)
out = ["# MFMA pipe utilisation per kernel (avg per launch): matrix-pipe cycles per SIMD / kernel cycles",
       f"{'kernel':90s} {'launches':>8s} {'kernel_cyc':>12s} {'mfma_cyc/SIMD':>14s} {'busy_%':>7s}"]
for _, k, n, cyc, busy, pct in rows:
    out.append(f"{k[:90]:90s} {n:8d} {cyc:12.0f} {busy:14.0f} {pct:7.1f}")
text = "\n".join(out) + "\n"
(open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout).write(text)
