#!/usr/bin/env python3
"""Copy the rocprofv3 summaries of gpurun_out/<tag>/ (made by tools/profile.sh on the GPU box) into
profiles/ and write profiles/<tag>_summary.md + profiles/<tag>_traffic.json (per-launch HBM bytes of
the scoring kernel from the PMC passes; bench.py reports it as roofline.traffic)."""
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
KERNEL = "k_score_point"
lines = ["# rocprofv3 summaries, round tag `%s`\n" % tag,
         "Commands: `tools/profile.sh %s` (rocprofv3 --kernel-trace --stats of `bench.py`, of "
         "`bench.py --workload sweep` and `--workload mc`; separate --pmc passes for hc and sweep).\n" % tag]
for name in ("hc", "sweep", "mc"):
    st = os.path.join(src, name, "%s_kernel_stats.csv" % name)
    if not os.path.exists(st):
        continue
    shutil.copy(st, os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, name)))
    lines.append("## %s — kernel stats (`%s_%s_kernel_stats.csv`)\n" % (name, tag, name))
    lines.append("| kernel | calls | avg ns | min | max | % |\n|---|---|---|---|---|---|")
    rows = list(csv.DictReader(open(st)))
    for r in rows[:6]:
        lines.append("| `%s` | %s | %.0f | %s | %s | %s |" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]),
                                                           r["MinNs"], r["MaxNs"], r["Percentage"]))
    bj = os.path.join(src, "%s.bench.json" % name)
    try:
        d = json.load(open(bj))
        shutil.copy(bj, os.path.join(dst, "%s_%s_bench.json" % (tag, name)))
        r = d["roofline"]
        prof = [float(x["AverageNs"]) for x in rows if KERNEL in x["Name"]]
        lines.append("\nrocprofv3 average for %s in the profiled run: %s us.  The profiled run's own bench line "
                     "(`%s_%s_bench.json`) is perturbed by the profiler (value %.4g %s, %.4f ms/step, attached HIP "
                     "events read %.2f us/launch)."
                     % (KERNEL, ", ".join("%.2f" % (p / 1e3) for p in prof), tag, name, d["value"], d["unit"],
                        d["ms_per_step"], r["avg_launch_us"]))
        pj = os.path.join(src, "%s.plain.json" % name)
        if os.path.exists(pj) and os.path.getsize(pj) > 2:
            pd_ = json.load(open(pj))
            shutil.copy(pj, os.path.join(dst, "%s_%s_bench_unprofiled.json" % (tag, name)))
            pr = pd_["roofline"]
            lines.append("Same command without the profiler (`%s_%s_bench_unprofiled.json`): value %.4g %s, %.4f "
                         "ms/step; HIP events attached to the dispatches: **%.2f us/launch** over %d launches -> "
                         "%.0f GB/s algorithmic = %.3f of 8 TB/s.\n"
                         % (tag, name, pd_["value"], pd_["unit"], pd_["ms_per_step"], pr["avg_launch_us"],
                            pr["launches"], pr["achieved"], pr["frac"]))
    except Exception as e:  # noqa: BLE001
        lines.append("\n(bench line not captured: %s)\n" % e)

traffic = {}
PMC_KERNEL = {"hc": "k_score_point", "sweep": "k_score_point", "pf": "k_score_gmapping"}
for wl in ("hc", "sweep", "pf"):
    for c in ("FETCH_SIZE", "WRITE_SIZE", "sq"):
        f = os.path.join(src, "pmc_%s_%s" % (wl, c), "pmc_counter_collection.csv")
        if not os.path.exists(f):
            continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if PMC_KERNEL[wl] in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        if not acc:
            continue
        lines.append("## PMC pass %s, workload %s (%s dispatches)\n" % (c, wl, PMC_KERNEL[wl]))
        for k, v in acc.items():
            lines.append("* %s: mean %.6g over %d dispatches" % (k, sum(v) / len(v), len(v)))
            if k in ("FETCH_SIZE", "WRITE_SIZE"):
                traffic.setdefault(wl, {})[k + "_kb_raw"] = sum(v) / len(v)
                traffic[wl][k + "_dispatches"] = len(v)
        lines.append("")
        with open(os.path.join(dst, "%s_pmc_%s_%s.csv" % (tag, wl, c)), "w") as out:
            out.write("counter,dispatches,mean\n")
            for k, v in acc.items():
                out.write("%s,%d,%.6g\n" % (k, len(v), sum(v) / len(v)))
for wl, t in traffic.items():
    if "FETCH_SIZE_kb_raw" in t and "WRITE_SIZE_kb_raw" in t:
        # MI355X_MICROARCH.md, HBM: rocprofv3 reports KB; on gfx950 FETCH_SIZE tallies 128-B
        # requests at 64 B -> x2; WRITE_SIZE taken as reported (uncalibrated, and tiny here)
        t["bytes_per_launch"] = 1024.0 * (2.0 * t["FETCH_SIZE_kb_raw"] + t["WRITE_SIZE_kb_raw"])
        t["correction"] = "FETCH_SIZE KB x2 (gfx950), WRITE_SIZE KB as reported; separate --pmc passes"
        t["kernel"] = PMC_KERNEL[wl]
        lines.append("* %s: HBM traffic per %s launch = %.0f bytes (%s)" % (wl, PMC_KERNEL[wl], t["bytes_per_launch"],
                                                                          t["correction"]))
if traffic:
    json.dump({"tag": tag, "kernels": PMC_KERNEL, "workloads": traffic},
              open(os.path.join(dst, "%s_traffic.json" % tag), "w"), indent=1)
open(os.path.join(dst, "%s_summary.md" % tag), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
