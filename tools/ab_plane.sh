#!/bin/bash
# A/B of the TBM probability plane (r06; SLAMHIP_OPT_TBM_PLANE) inside ONE gpurun call: the Monte-Carlo headline workload
# (cfg3) and the vinySLAM world loop with the plane on and off.   tools/ab_plane.sh [steps]
mkdir -p gpurun_out/r06
steps=${1:-40}
for rep in 1 2; do
for pl in 1 0; do
  timeout 300 python bench.py --workload mc --legs none --no-cpu --tbm-plane $pl --steps $steps --detail-out gpurun_out/r06/mc_plane$pl.json > /dev/null 2> gpurun_out/r06/mc_plane$pl.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r06/mc_plane$pl.json"))
    c = d["config"]
    print("plane $pl: %.4f ms/step (resident scan %.4f), %.3e units/s, frac %.3f, %.1f us/launch, super-steps %.1f"
          % (d["ms_per_step"], c.get("ms_per_step_resident", float("nan")), d["value"], d["roofline"]["frac"], d["roofline"].get("avg_launch_us"),
             c.get("super_steps_per_match")))
except Exception as e:
    print("plane $pl: no record (%s)" % e)
PY
done
done
for pl in 1 0; do
  timeout 300 python bench.py --legs world_viny,mc --no-cpu --tbm-plane $pl --steps 5 --warmup 2 --detail-out gpurun_out/r06/wv_plane$pl.json > /dev/null 2> gpurun_out/r06/wv_plane$pl.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r06/wv_plane$pl.json"))
    w, m = d["world_loop_viny"], d["monte_carlo"]
    print("plane $pl: vinySLAM world loop %.4f ms/scan; mc leg %.4f ms/step, frac %.3f, %.1f us/launch, gather %.0f GB/s"
          % (w["ms_per_scan"], m["ms_per_step"], m["roofline"]["frac"], m["roofline"]["avg_launch_us"], m["roofline"]["achieved_gather_gbs"]))
except Exception as e:
    print("plane $pl: no record (%s)" % e)
PY
done
