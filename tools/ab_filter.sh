#!/bin/bash
# A/B of the filter legs (pf: likelihood step of 100 chains; pf_update: the shared-map step, one chain per particle
# after the other) with the chains as a kernel per super-step (0) / one co-resident launch (1); run on the GPU box
mkdir -p gpurun_out/r05
for rc in 0 1; do
  timeout 600 python bench.py --legs pf,pf_update --no-cpu --steps 8 --resident-chains $rc > gpurun_out/r05/filter_rc$rc.json 2> gpurun_out/r05/filter_rc$rc.err
  python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/r05/filter_rc$rc.json").read().splitlines() if l.startswith("{")][-1])
    pf = d["particle_filter"]
    print("resident chains $rc: pf %.3f ms/step (%.0f particles/s), launches last step %s; pf_update %s" % (
        pf.get("ms_per_step", float("nan")), pf.get("value", float("nan")), pf.get("launches_last_step"),
        json.dumps({k: pf.get("with_map_update", {}).get(k) for k in ("ms_per_step", "value")})))
except Exception as e:
    print("resident chains $rc: no line (%s)" % e)
PY
  tail -2 gpurun_out/r05/filter_rc$rc.err
done
