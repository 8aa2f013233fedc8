#!/usr/bin/env python3
"""Average duration of the K6 (map update) kernels in a rocprofv3 kernel trace, split into the
single-scan launches (one particle at a time) and the batch launches (all particles at once).
usage: k6_trace_summary.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if not any(k in n for k in ("mu_", "rocprim", "tile", "table")):
        continue
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    g = int(r["Grid_Size_X"])
    per_beam = any(k in n for k in ("k_mu_emit", "k_mu_count", "k_mu_beam_ids"))
    big = g > (20_000 if per_beam else 2_000_000)
    short = n.split("(")[0]
    if "rocprim" in short:
        short = "rocprim::" + ("onesweep" if "onesweep" in n else "merge" if "merge" in n else "block_sort" if "block_sort" in n else "scan")
    agg[(short[-48:], big)].append(d)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k[0]:50s} {'batch ' if k[1] else 'single'} calls {len(v):5d} avg {sum(v)/len(v)/1e3:9.1f} us total {sum(v)/1e6:8.2f} ms")
