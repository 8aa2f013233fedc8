#!/usr/bin/env python3
"""ms per step of the secondary legs of a bench.py line read from stdin (pf, pf_update, pf_maps, cfg5, world) -- a
helper for A/B runs on the GPU box: python bench.py --legs pf_update ... | python tools/leg_ms.py"""
import json
import sys

line = [ln for ln in sys.stdin.read().splitlines() if ln.startswith('{"metric')][-1]
d = json.loads(line)
out = {"headline_ms": round(d["ms_per_step"], 4)}
pf = d.get("particle_filter") or {}
if "ms_per_step" in pf:
    out["pf_ms"] = round(pf["ms_per_step"], 4)
    out["pf_launch_us"] = round((pf.get("roofline") or {}).get("avg_launch_us", 0.0), 2)
for k, name in (("with_map_update", "pf_update"), ("with_particle_maps", "pf_maps")):
    if k in pf and "ms_per_step" in pf[k]:
        out[name + "_ms"] = round(pf[k]["ms_per_step"], 3)
        r = pf[k].get("roofline_map_update") or {}
        if "avg_launch_us" in r:
            out[name + "_k6_us"] = round(r["avg_launch_us"], 1)
c5 = d.get("cfg5") or {}
if "ms_per_step" in c5:
    out["cfg5_ms"] = round(c5["ms_per_step"], 3)
    out["cfg5_k6_us"] = round(c5["roofline"]["avg_launch_us"], 0)
    out["cfg5_k3_us"] = round(c5["roofline_likelihood"]["avg_launch_us"], 1)
w = d.get("world_loop") or {}
if "ms_per_scan" in w:
    out["world_ms"] = round(w["ms_per_scan"], 4)
print(out)
