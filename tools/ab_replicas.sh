#!/bin/bash
# A/B of the `replicas` leg (K matches per call) under the two device-chain forms; run on the GPU box through gpurun
mkdir -p gpurun_out/r05
for cm in 1 2; do
  timeout 400 python bench.py --legs replicas --no-cpu --chain-mode $cm --steps 64 > gpurun_out/r05/rep_cm$cm.json 2> gpurun_out/r05/rep_cm$cm.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r05/rep_cm$cm.json").read().strip().splitlines()[-1])
    print("chain mode $cm: headline %.4f ms/step" % d["ms_per_step"])
    for r in d["replicas"]["by_K"]:
        print("  K %2d: %.3f ms/call, %.2f G units/s, frac %.3f, %.1f us/launch, kernels/call %s, spec %.2f, resident %s/%s" % (
            r["K"], r["ms_per_call"], r["value"] / 1e9, r["roofline"]["frac"], r["roofline"]["avg_launch_us"],
            r["kernels_per_call"], r["speculation_ratio"], r.get("co_resident_launches"), r.get("co_resident_gave_up")))
except Exception as e:
    print("chain mode $cm: no line (%s)" % e)
PY
  tail -2 gpurun_out/r05/rep_cm$cm.err
done
