#!/usr/bin/env python3
"""Timeline of one particle-filter step with per-particle maps from a rocprofv3 kernel trace: the
kernels between the last two batch k_mu_count launches, with the idle gaps in front of them.
usage: step_timeline.py <kernel_trace.csv> [min_us]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_mu_count" in r["Kernel_Name"] and int(r["Grid_Size_X"]) > 20000]
a, b = idx[-2], idx[-1]
seg = rows[a:b]
t0 = int(seg[0]["Start_Timestamp"])
t1 = int(rows[b]["Start_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
print(f"step wall {(t1 - t0) / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us, launches {len(seg)}")
prev_end = None
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    name = r["Kernel_Name"].split("(")[0][-44:]
    if gap > min_us or (e - s) / 1e3 > min_us:
        print(f"{(s - t0) / 1e3:8.1f} us  gap {gap:7.1f}  run {(e - s) / 1e3:7.1f}  {name}")
    prev_end = e
