#!/usr/bin/env python3
"""Mean of every collected counter per kernel (name prefix) from a rocprofv3 --pmc counter_collection csv.
usage: pmc_by_kernel.py <pmc_counter_collection.csv> [name-substring ...]"""
import collections
import csv
import sys

want = sys.argv[2:]
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0][-48:]
    if want and not any(w in r["Kernel_Name"] for w in want):
        continue
    k = (name, r["Grid_Size"] if "Grid_Size" in r else "", r["Counter_Name"])
    acc[k][0] += float(r["Counter_Value"])
    acc[k][1] += 1
for (name, grid, ctr), (s, n) in sorted(acc.items()):
    if n >= int(__import__("os").environ.get("PMC_MIN_N", "3")):
        print(f"{name:50s} grid {grid:>9s} {ctr:22s} mean {s / n:14.1f} over {n}")
