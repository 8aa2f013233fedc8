"""A/B of SLAMHIP_EXP values on one box (temporary): resident cfg2 step, interleaved rounds."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402
from bench_legs.common import WORKLOADS, rotating_scenes  # noqa: E402
from synth import make_scene  # noqa: E402

pkg = ge.load_package()
cell, weighting, kind, params, _, _ = WORKLOADS["hc"]
sc = make_scene(cell_model=cell, size=2000, scale=0.05, n_beams=1080, seed=100, weighting=weighting)
scenes = rotating_scenes(sc, 1080, weighting)
ctx = pkg.Context(0)
ctx.upload_map(0, sc["map"])
m = pkg.Matcher(ctx, kind, pkg.spe_cfg(), params)
for j, s_ in enumerate(scenes):
    c_, s__ = pkg.beam_trig(s_["angle"])
    ctx.scan_store(j, s_["range"], c_, s__, s_["weight"])
vals = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0,1,2,3").split(",")]
N = 320
ref = {}
for v in vals:
    os.environ["SLAMHIP_EXP"] = str(v)
    for k in range(16):
        ctx.scan_select(k)
        r = m.process_scan(0, scenes[k]["init_pose"], trace=True)
        key = (r["prob"], tuple(r["delta"]), r["n_calls"], tuple(r["scores"]))
        if k in ref:
            assert ref[k] == key, ("exp %d differs on scene %d" % (v, k))
        else:
            ref[k] = key
print("results identical over", vals)
res = {v: [] for v in vals}
pc = time.perf_counter
for rnd in range(6):
    for v in vals:
        os.environ["SLAMHIP_EXP"] = str(v)
        for i in range(32):
            ctx.scan_select(i % 16)
            m.process_scan(0, scenes[i % 16]["init_pose"])
        t0 = pc()
        for i in range(N):
            k = i % 16
            ctx.scan_select(k)
            m.process_scan(0, scenes[k]["init_pose"])
        res[v].append(1e6 * (pc() - t0) / N)
for v in vals:
    print("exp %d: resident step us: %s  median %.2f" % (v, " ".join("%.2f" % x for x in res[v]), float(np.median(res[v]))))
print(m.resident_stats())
