#!/usr/bin/env python3
"""Per-scene view of the headline's rotating set: ms per match (5 repeats), scorer calls, poses evaluated, kernels
launched, super-steps, re-scored steps -- to see which scenes make the slow matches."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.argv = ["bench.py"]
import bench
import __graft_entry__ as ge
from synth import make_scene
pkg = ge.load_package()
sc = make_scene(cell_model=0, size=2000, scale=0.05, n_beams=1080, seed=100)
scenes = bench.rotating_scenes(sc, 1080, "even")
ctx = pkg.Context(0)
ctx.upload_map(0, sc["map"])
for j, s in enumerate(scenes):
    c, sn = pkg.beam_trig(s["angle"])
    ctx.scan_store(j, s["range"], c, sn, s["weight"])
m = pkg.Matcher(ctx, "HC", pkg.spe_cfg(), [128, 0.1, 0.1])
for rep in range(3):
    for j, s in enumerate(scenes):
        ctx.scan_select(j)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            r = m.process_scan(0, s["init_pose"])
            ts.append(1e3 * (time.perf_counter() - t0))
        st = m.stats()
        if rep == 2:
            print("scene %2d err %.1f beams %d: ms %s calls %d evaluated %d steps %d kernels %d rescored %d prob %.4f"
                  % (j, s["error_x_default"], s["range"].size, " ".join("%.3f" % t for t in ts), st["scorer_calls"],
                     st["poses_evaluated"], st["launches"], st["kernels_launched"], st["steps_rescored"], r["prob"]))
# rotating order, one pass
ts = []
for i in range(64):
    j = i % len(scenes)
    ctx.scan_select(j)
    t0 = time.perf_counter()
    m.process_scan(0, scenes[j]["init_pose"])
    ts.append(1e3 * (time.perf_counter() - t0))
print("rotating:", " ".join("%.3f" % t for t in ts))
