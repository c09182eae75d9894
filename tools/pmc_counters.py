#!/usr/bin/env python
"""Average value per launch of every counter in a rocprofv3 --pmc counter_collection.csv, per kernel (top kernels by launches x value of
the first counter).  usage: pmc_counters.py <counter_collection.csv> [kernel substring ...]"""
import collections
import csv
import sys

path, subs = sys.argv[1], sys.argv[2:]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(path)):
    k = r["Kernel_Name"]
    if subs and not any(s in k for s in subs):
        continue
    tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[k][r["Counter_Name"]] += 1
for k in tot:
    print(k[:110])
    for c in sorted(tot[k]):
        print(f"    {c:44s} {tot[k][c] / cnt[k][c]:16.1f}   (x{cnt[k][c]})")
