#!/bin/bash
# A/B of the Monte-Carlo workload (BASELINE configs[2]) by candidates per super-step of the co-resident launch
# (slamhip_matcher_set_batch: 384 = r04's grid of 385 workgroups, 511 = r05's 512), inside ONE gpurun call:
#   tools/ab_mc_slots.sh [steps] [tag]
tag=${2:-r05}
mkdir -p gpurun_out/$tag
steps=${1:-60}
for slots in 384 511 384 511; do
  timeout 300 python bench.py --workload mc --legs none --no-cpu --batch $slots --steps $steps > gpurun_out/$tag/mc_slots$slots.json 2> gpurun_out/$tag/mc_slots$slots.err
  python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/$tag/mc_slots$slots.json").read().splitlines() if l.startswith("{")][-1])
    c = d["config"]
    print("slots $slots: %.4f ms/step (resident scan %.4f), %.3e units/s, frac %.3f, %s us/launch, super-steps %.1f, spec %.2f, resident %s"
          % (d["ms_per_step"], c.get("ms_per_step_resident", float("nan")), d["value"], d["roofline"]["frac"], d["roofline"].get("avg_launch_us"),
             c.get("super_steps_per_match"), c.get("speculation_ratio"), c.get("resident")))
except Exception as e:
    print("slots $slots: no line (%s)" % e)
PY
  tail -2 gpurun_out/$tag/mc_slots$slots.err | grep -v amdgpu.ids
done
