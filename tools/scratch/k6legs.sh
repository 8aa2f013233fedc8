#!/bin/bash
# bench the three K6-heavy legs and print their ms/step (scratch tool)
for leg in pf_update pf_maps cfg5; do
  timeout 300 python bench.py --legs $leg --no-cpu --steps 20 2>/dev/null | tail -1 > /tmp/k6leg.json
  python3 - <<'PY'
import json
d=json.load(open('/tmp/k6leg.json')); p=dict(d.get("particle_filter") or {}); p["cfg5"]=d.get("cfg5")
for k,v in p.items():
    if isinstance(v,dict) and "ms_per_step" in v:
        r=v.get("roofline_map_update") or v.get("roofline") or {}
        print(k, round(v["value"],1), round(v["ms_per_step"],3), "K6 us", r.get("avg_launch_us"), {kk:r.get(kk) for kk in ("achieved",)})
PY
done
