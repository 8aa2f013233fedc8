#!/usr/bin/env python3
"""Copy the rocprofv3 summaries of gpurun_out/<tag>/ (made by tools/profile.sh on the GPU box) into
profiles/ and write profiles/<tag>_summary.md + profiles/<tag>_traffic.json: per launch of each workload's
dominant kernel the HBM bytes (FETCH_SIZE / WRITE_SIZE passes) and the VALU issue figures (SQ pass), which
bench.py reports as roofline.traffic / roofline_valu."""
import collections
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
CLOCK_GHZ = 2.4  # MI355X peak engine clock (MI355X_MICROARCH.md); SQ_BUSY_CYCLES / duration is printed beside it
# dominant kernel of each leg (substring of the rocprof kernel name)
LEG_KERNEL = {"hc": "k_hc_chain_resident", "sweep": "k_score_point", "mc": "k_mc_chain_resident", "pf": "k_hc_chain_resident_gm",
              "pf_update": "k_hc_chain_step", "pf_maps": "k_mu_", "cfg5": "k_mu_", "world": "k_hc_chain_resident",
              "replicas": "k_hc_chain_resident", "bf": "k_score_point", "mc_leg": "k_mc_chain_resident", "world_viny": "k_mu_cells"}
lines = ["# rocprofv3 summaries, round tag `%s`\n" % tag,
         "Commands: `tools/profile.sh %s` -- one `rocprofv3 --kernel-trace --stats` run per leg of `bench.py` "
         "(`--legs none` = the headline alone, `--workload sweep`, `--workload mc`, `--legs pf`, `pf_update`, `pf_maps`, "
         "`cfg5`), the same commands without the profiler, and separate `--pmc` passes.\n" % tag]


def short(name):
    return name.replace("void ", "").replace("slamhip::", "").replace("(anonymous namespace)::", "")[:72]


for name in ("hc", "sweep", "mc", "pf", "pf_update", "pf_maps", "cfg5", "world", "replicas", "bf", "mc_leg", "world_viny"):
    st = os.path.join(src, name, "%s_kernel_stats.csv" % name)
    if not os.path.exists(st):
        continue
    shutil.copy(st, os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, name)))
    lines.append("## %s -- kernel stats (`%s_%s_kernel_stats.csv`)\n" % (name, tag, name))
    lines.append("| kernel | calls | avg us | min | max | % |\n|---|---|---|---|---|---|")
    rows = list(csv.DictReader(open(st)))
    for r in rows[:8]:
        lines.append("| `%s` | %s | %.2f | %.2f | %.2f | %s |" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3,
                                                              float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
    bj = os.path.join(src, "%s.bench.json" % name)
    try:
        d = json.load(open(bj))
        shutil.copy(bj, os.path.join(dst, "%s_%s_bench.json" % (tag, name)))
        pj = os.path.join(src, "%s.plain.json" % name)
        if os.path.exists(pj) and os.path.getsize(pj) > 2:
            pd_ = json.load(open(pj))
            shutil.copy(pj, os.path.join(dst, "%s_%s_bench_unprofiled.json" % (tag, name)))
            if name in ("hc", "sweep", "mc"):
                r, pr = d["roofline"], pd_["roofline"]
                prof = [float(x["AverageNs"]) for x in rows if pr["kernel"] in x["Name"]]
                lines.append("\n%s: rocprofv3 average %s us; HIP events attached to the dispatches of the un-profiled run "
                             "(`%s_%s_bench_unprofiled.json`): **%.2f us/launch** over %d launches -> %.0f GB/s algorithmic = "
                             "%.3f of 8 TB/s; %.4f ms/step, value %.4g %s.  (Profiled run: %.4f ms/step.)\n"
                             % (pr["kernel"], ", ".join("%.2f" % (p / 1e3) for p in prof), tag, name, pr["avg_launch_us"],
                                pr["launches"], pr["achieved"], pr["frac"], pd_["ms_per_step"], pd_["value"], pd_["unit"],
                                d["ms_per_step"]))
            else:
                leg = {"pf": None, "pf_update": "with_map_update", "pf_maps": "with_particle_maps"}.get(name, name)
                if name in ("cfg5", "world", "replicas", "bf", "mc_leg", "world_viny"):
                    obj = pd_.get({"world": "world_loop", "bf": "brute_force", "mc_leg": "monte_carlo",
                                   "world_viny": "world_loop_viny"}.get(name, name), {})
                    if name == "replicas":
                        lines.append("\nun-profiled replicas leg (`%s_%s_bench_unprofiled.json`): " % (tag, name) + "; ".join(
                            "K=%d %.3f ms/call = %.3g units/s (kernel frac %.3f)" %
                            (r_["K"], r_["ms_per_call"], r_["value"], r_["roofline"]["frac"])
                            for r_ in obj.get("by_K", [])) + "\n")
                        obj = {}
                else:
                    obj = pd_.get("particle_filter", {})
                    if leg:
                        obj = obj.get(leg, {})
                if obj:
                    lines.append("\nun-profiled line of this leg (`%s_%s_bench_unprofiled.json`): %.3f ms/step = %.0f %s\n"
                                 % (tag, name, obj.get("ms_per_step", obj.get("ms_per_match", obj.get("ms_per_scan", float("nan")))),
                                    obj.get("value", float("nan")), obj.get("unit", "")))
    except Exception as e:  # noqa: BLE001
        lines.append("\n(bench line not captured: %s)\n" % e)
dj = os.path.join(src, "default.plain.json")
if os.path.exists(dj) and os.path.getsize(dj) > 2:
    shutil.copy(dj, os.path.join(dst, "%s_default_bench_unprofiled.json" % tag))
    lines.append("The driver's command (`python bench.py --steps 20 --warmup 5`, every leg, CPU baselines): full record "
                 "(the `--detail-out` sidecar) `%s_default_bench_unprofiled.json`.\n" % tag)
lj = os.path.join(src, "default.line.json")
if os.path.exists(lj) and os.path.getsize(lj) > 2:
    shutil.copy(lj, os.path.join(dst, "%s_default_bench_line.json" % tag))
    lines.append("... and the LAST stdout line of that run, what the driver parses (%d bytes): `%s_default_bench_line.json`.\n"
                 % (os.path.getsize(lj), tag))

rs_ = os.path.join(src, "resident_stamps.txt")
if os.path.exists(rs_):
    shutil.copy(rs_, os.path.join(dst, "%s_resident_stamps.txt" % tag))
    lines.append("In-kernel timeline of the co-resident chain's super-step (`tools/hc_resident_stamps.py`): `%s_resident_stamps.txt`.\n" % tag)
bs_ = os.path.join(src, "batch_stamps.txt")
if os.path.exists(bs_):
    shutil.copy(bs_, os.path.join(dst, "%s_batch_stamps.txt" % tag))
    lines.append("In-kernel timeline of chain 0 of a co-resident batch (`tools/hc_batch_stamps.py`): `%s_batch_stamps.txt`.\n" % tag)
cs = os.path.join(src, "chain_stamps.txt")
if os.path.exists(cs):
    shutil.copy(cs, os.path.join(dst, "%s_chain_stamps.txt" % tag))
    lines.append("In-kernel timeline of the hill-climbing chain's super-step (`tools/hc_chain_stamps.py`): `%s_chain_stamps.txt`.\n" % tag)

traffic = {}
PMC_KERNEL = {"hc": "k_hc_chain_resident<0", "sweep": "k_score_point", "mc": "k_mc_chain_resident", "pf": "k_hc_chain_resident_gm"}
for wl in ("hc", "sweep", "mc", "pf"):
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
        # kernel duration inside the same pass (the PMC run's own kernel trace)
        dur = []
        kt = os.path.join(src, "pmc_%s_%s" % (wl, c), "pmc_kernel_trace.csv")
        if os.path.exists(kt):
            for r in csv.DictReader(open(kt)):
                if PMC_KERNEL[wl] in r["Kernel_Name"]:
                    dur.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
        lines.append("## PMC pass %s, workload %s (%s dispatches)\n" % (c, wl, PMC_KERNEL[wl]))
        with open(os.path.join(dst, "%s_pmc_%s_%s.csv" % (tag, wl, c)), "w") as out:
            out.write("counter,dispatches,mean\n")
            for k, v in acc.items():
                out.write("%s,%d,%.6g\n" % (k, len(v), sum(v) / len(v)))
                lines.append("* %s: mean %.6g over %d dispatches" % (k, sum(v) / len(v), len(v)))
                if k in ("FETCH_SIZE", "WRITE_SIZE"):
                    traffic.setdefault(wl, {})[k + "_kb_raw"] = sum(v) / len(v)
                    traffic[wl][k + "_dispatches"] = len(v)
        if c == "sq" and "SQ_INSTS_VALU" in acc:
            insts = sum(acc["SQ_INSTS_VALU"]) / len(acc["SQ_INSTS_VALU"])
            t = traffic.setdefault(wl, {})
            v = {"SQ_INSTS_VALU": insts, "clock_ghz": CLOCK_GHZ}
            if dur:
                mean_ns = sum(dur) / len(dur)
                v["kernel_ns_in_pmc_pass"] = mean_ns
                # 1024 SIMDs, one VALU wave-instruction per 4 cycles each
                v["issue_frac"] = insts / (1024.0 * mean_ns * CLOCK_GHZ / 4.0)
                v["note"] = ("SQ_INSTS_VALU per dispatch / (1024 SIMDs x duration x %.1f GHz / 4 cycles); duration = this "
                             "kernel's mean in the PMC pass's own kernel trace" % CLOCK_GHZ)
                lines.append("* VALU issue fraction: %.3f (%.0f wave-instructions per dispatch, %.2f us)" %
                             (v["issue_frac"], insts, mean_ns / 1e3))
            t["valu"] = v
        lines.append("")
# K6: HBM bytes per map-update pipeline = the counters of ALL its dispatches (k_mu_*, rocprim scan / sort) added up,
# divided by the number of pipelines (one k_mu_apply each)
for wl in ("pf_update", "pf_maps", "cfg5", "world"):
    t = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = os.path.join(src, "pmc_%s_%s" % (wl, c), "pmc_counter_collection.csv")
        if not os.path.exists(f):
            continue
        total, pipelines = 0.0, 0
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if r["Counter_Name"] != c or not ("k_mu_" in name or "rocprim" in name):
                continue
            total += float(r["Counter_Value"])
            pipelines += 1 if ("k_mu_apply" in name or "k_mu_cells" in name) else 0
        if pipelines:
            t[c + "_kb_raw"] = total / pipelines
            t["pipelines"] = pipelines
    if "FETCH_SIZE_kb_raw" in t and "WRITE_SIZE_kb_raw" in t:
        t["kernel"] = "K6 pipeline (k_mu_* and rocprim dispatches of one map update)"
        t["bytes_per_launch"] = 1024.0 * (2.0 * t["FETCH_SIZE_kb_raw"] + t["WRITE_SIZE_kb_raw"])
        t["correction"] = "FETCH_SIZE KB x2 (gfx950), WRITE_SIZE KB as reported; separate --pmc passes; summed over the pipeline's dispatches"
        traffic["k6_" + wl] = t
        lines.append("## PMC, K6 pipeline of leg %s\n\n* HBM traffic per map-update pipeline = %.0f bytes over %d pipelines (%s)\n"
                     % (wl, t["bytes_per_launch"], t["pipelines"], t["correction"]))
# units per launch (for instructions per unit) from the un-profiled bench lines
for wl in [w for w in traffic if not w.startswith("k6_")]:
    pj = os.path.join(src, "%s.plain.json" % wl)
    try:
        if os.path.exists(pj) and os.path.getsize(pj) > 0:
            pd_ = json.load(open(pj))
        else:
            # on the GPU box the PMC passes come first: the SQ pass's own bench line has the same launch shape
            pd_ = [json.loads(ln) for ln in open(os.path.join(src, "pmc_%s_sq.log" % wl)) if ln.startswith('{"metric')][-1]
        r = pd_["particle_filter"]["roofline"] if wl == "pf" else pd_["roofline"]
        upl = r["units_launched"] / max(r["launches"], 1)
        if "valu" in traffic[wl]:
            traffic[wl]["valu"]["units_per_launch"] = upl
            traffic[wl]["valu"]["insts_per_unit"] = traffic[wl]["valu"]["SQ_INSTS_VALU"] * 64.0 / upl
    except Exception:  # noqa: BLE001
        pass
for wl, t in traffic.items():
    if wl.startswith("k6_"):
        continue
    t["kernel"] = PMC_KERNEL[wl]
    if "FETCH_SIZE_kb_raw" in t and "WRITE_SIZE_kb_raw" in t:
        # MI355X_MICROARCH.md, HBM: rocprofv3 reports KB; on gfx950 FETCH_SIZE tallies 128-B
        # requests at 64 B -> x2; WRITE_SIZE taken as reported
        t["bytes_per_launch"] = 1024.0 * (2.0 * t["FETCH_SIZE_kb_raw"] + t["WRITE_SIZE_kb_raw"])
        t["correction"] = "FETCH_SIZE KB x2 (gfx950), WRITE_SIZE KB as reported; separate --pmc passes"
        lines.append("* %s: HBM traffic per %s launch = %.0f bytes (%s)" % (wl, PMC_KERNEL[wl], t["bytes_per_launch"],
                                                                          t["correction"]))
if traffic:
    json.dump({"tag": tag, "kernels": PMC_KERNEL, "workloads": traffic},
              open(os.path.join(dst, "%s_traffic.json" % tag), "w"), indent=1)
open(os.path.join(dst, "%s_summary.md" % tag), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
