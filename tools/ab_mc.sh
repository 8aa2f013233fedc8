#!/bin/bash
# A/B of the Monte-Carlo workload (BASELINE configs[2]: vinySLAM MC, 4096 candidates, TBM cells) under the two
# device-chain forms (run on the GPU box through gpurun):  tools/ab_mc.sh [steps]
mkdir -p gpurun_out/r04
steps=${1:-60}
for cm in 1 2; do
  timeout 300 python bench.py --workload mc --legs none --no-cpu --chain-mode $cm --steps $steps > gpurun_out/r04/mc_cm$cm.json 2> gpurun_out/r04/mc_cm$cm.err
  python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/r04/mc_cm$cm.json").read().splitlines() if l.startswith("{")][-1])
    c = d["config"]
    print("chain mode $cm: %.4f ms/step (resident scan %.4f), %.3e units/s, frac %.3f, %s us/launch, super-steps %.1f, spec %.2f, resident %s, parity %s"
          % (d["ms_per_step"], c.get("ms_per_step_resident", float("nan")), d["value"], d["roofline"]["frac"], d["roofline"].get("avg_launch_us"),
             c.get("super_steps_per_match"), c.get("speculation_ratio"), c.get("resident"), (d.get("parity") or {}).get("traces_equal")))
except Exception as e:
    print("chain mode $cm: no line (%s)" % e)
PY
  tail -2 gpurun_out/r04/mc_cm$cm.err | grep -v amdgpu.ids
done
