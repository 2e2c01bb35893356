#!/usr/bin/env python3
"""Averages rocprofv3 --pmc counter_collection CSV rows per (kernel, counter)."""
import csv, sys, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"][:60], r["Counter_Name"])
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for (k, c), (s, n) in sorted(acc.items()):
    if any(t in k for t in ("k1_", "k12_", "k2", "k_")):
        print(f"{k:60s} {c:24s} avg {s / n:16.1f} over {n}")
