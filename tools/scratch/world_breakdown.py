"""The tinySLAM world loop of bench_legs.single_hypothesis.world_leg split into its three calls (diagnostic)."""
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
m0 = sc["map"]
trig = [pkg.beam_trig(s["angle"]) for s in scenes]
ctx.map_bind(5, m0.cell_model, m0.width, m0.height, m0.origin, m0.scale, m0.unknown)
ctx.map_upload_window(5, 0, 0, m0.payload)
ctx.map_set_auto_grow(5, True)
m = pkg.Matcher(ctx, kind, pkg.spe_cfg(), params)
ctx.map_set_deferred(True)
pc = time.perf_counter
tu = tm = ta = 0.0
N = 200
for i in range(N + 32):
    k = i % 16
    s, (cos_a, sin_a) = scenes[k], trig[k]
    a = pc()
    ctx.scan_upload(s["range"], cos_a, sin_a, s["weight"], None)
    b = pc()
    r = m.process_scan(5, s["init_pose"])
    c = pc()
    ctx.map_append_scan(5, pkg.RULE_MEAN, s["init_pose"] + r["delta"], s["range"], cos_a, sin_a)
    d = pc()
    if i >= 32:
        tu += b - a
        tm += c - b
        ta += d - c
ctx.map_drain()
print("per scan: scan_upload %.1f us, process_scan %.1f us, map_append_scan %.1f us (queued), total %.1f" %
      (1e6 * tu / N, 1e6 * tm / N, 1e6 * ta / N, 1e6 * (tu + tm + ta) / N))
