#!/bin/bash
# A/B of the hill-climbing headline under the two device-chain forms (run on the GPU box through gpurun):
#   tools/ab_chain.sh [steps]   -> gpurun_out/r04/hc_cm{1,2}.json + one summary line each
mkdir -p gpurun_out/r04
steps=${1:-200}
for cm in 1 2; do
  timeout 300 python bench.py --legs none --no-cpu --chain-mode $cm --steps $steps > gpurun_out/r04/hc_cm$cm.json 2> gpurun_out/r04/hc_cm$cm.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r04/hc_cm$cm.json").read().strip().splitlines()[-1])
    c = d["config"]
    print("chain mode $cm: %.4f ms/step, %.3e units/s, roofline frac %.3f, launches/step %s, resident %s, ms/match %s, busy %s"
          % (d["ms_per_step"], d["value"], d["roofline"]["frac"], c.get("launches_per_step"), c.get("resident"),
             c.get("ms_per_match"), c.get("kernel_busy_frac")))
except Exception as e:
    print("chain mode $cm: no line (%s)" % e)
PY
  tail -3 gpurun_out/r04/hc_cm$cm.err
done
