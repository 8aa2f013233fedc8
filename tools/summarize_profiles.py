#!/usr/bin/env python3
"""Copy the rocprofv3 summaries of gpurun_out/<tag>/ (made by tools/profile.sh on the GPU box) into
profiles/ and write profiles/<tag>_summary.md."""
import collections
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
lines = ["# rocprofv3 summaries, round tag `%s`\n" % tag,
         "Commands: `tools/profile.sh %s` (rocprofv3 --kernel-trace --stats of `bench.py`, of "
         "`bench.py --workload sweep` and `--workload mc`; separate --pmc passes).\n" % tag]
for name in ("hc", "sweep", "mc"):
    st = os.path.join(src, name, "%s_kernel_stats.csv" % name)
    if not os.path.exists(st):
        continue
    shutil.copy(st, os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, name)))
    lines.append("## %s — kernel stats (`%s_%s_kernel_stats.csv`)\n" % (name, tag, name))
    lines.append("| kernel | calls | avg ns | min | max | % |\n|---|---|---|---|---|---|")
    for r in list(csv.DictReader(open(st)))[:4]:
        lines.append("| `%s` | %s | %.0f | %s | %s | %s |" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]),
                                                           r["MinNs"], r["MaxNs"], r["Percentage"]))
    bj = os.path.join(src, "%s.bench.json" % name)
    try:
        d = json.load(open(bj))
        shutil.copy(bj, os.path.join(dst, "%s_%s_bench.json" % (tag, name)))
        r = d["roofline"]
        lines.append("\nbench line of the same (profiled) run: value %.4g %s, %.4f ms/step; HIP-event kernel "
                     "time %.2f us/launch over %d launches -> %.0f GB/s algorithmic = %.3f of 8 TB/s.\n"
                     % (d["value"], d["unit"], d["ms_per_step"], r["avg_launch_us"], r["launches"],
                        r["achieved"], r["frac"]))
    except Exception as e:  # noqa: BLE001
        lines.append("\n(bench line not captured: %s)\n" % e)
for c in ("FETCH_SIZE", "WRITE_SIZE", "sq"):
    f = os.path.join(src, "pmc_%s" % c, "pmc_counter_collection.csv")
    if not os.path.exists(f):
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_score_point" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    lines.append("## PMC pass %s (sweep: 4096 poses x 1080 beams per launch, k_score_point)\n" % c)
    for k, v in acc.items():
        lines.append("* %s: mean %.6g over %d dispatches" % (k, sum(v) / len(v), len(v)))
    lines.append("")
    with open(os.path.join(dst, "%s_pmc_%s.csv" % (tag, c)), "w") as out:
        out.write("counter,dispatches,mean\n")
        for k, v in acc.items():
            out.write("%s,%d,%.6g\n" % (k, len(v), sum(v) / len(v)))
open(os.path.join(dst, "%s_summary.md" % tag), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
