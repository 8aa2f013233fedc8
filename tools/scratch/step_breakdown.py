"""Where a headline step's time goes on the host side (diagnostic, not a test): the bench's cfg2 raw-scan step split
into its two C calls, next to the resident form and an empty C call.  python tools/scratch/step_breakdown.py"""
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
raw = [ctx.make_raw_scan(0, s_["raw_range"], s_["raw_angle"], is_occ=s_["is_occ"], weighting=weighting) for s_ in scenes]
N = int(os.environ.get("N", "400"))
for k in range(3 * len(scenes)):
    raw[k % len(scenes)](scenes[k % len(scenes)]["init_pose"])
    m.process_scan(0, scenes[k % len(scenes)]["init_pose"])
ctx.synchronize()
pc = time.perf_counter
tu = tp = 0.0
t0 = pc()
for i in range(N):
    k = i % len(scenes)
    a = pc()
    raw[k](scenes[k]["init_pose"])
    b = pc()
    m.process_scan(0, scenes[k]["init_pose"])
    c = pc()
    tu += b - a
    tp += c - b
tot = pc() - t0
print("raw step: %.2f us = filter_upload %.2f + process_scan %.2f (+ %.2f loop/timer)" % (1e6 * tot / N, 1e6 * tu / N, 1e6 * tp / N,
                                                                                       1e6 * (tot - tu - tp) / N))
t0 = pc()
for i in range(N):
    k = i % len(scenes)
    ctx.scan_select(k)
    m.process_scan(0, scenes[k]["init_pose"])
print("resident step: %.2f us" % (1e6 * (pc() - t0) / N))
t0 = pc()
for i in range(N):
    k = i % len(scenes)
    raw[k](scenes[k]["init_pose"])
ctx.synchronize()
print("filter_upload alone, back to back: %.2f us" % (1e6 * (pc() - t0) / N))
t0 = pc()
for i in range(N):
    pkg.libm_variant()
print("an empty C call through ctypes: %.2f us" % (1e6 * (pc() - t0) / N))
st = m.stats()
print("stats of the last match:", {k_: st[k_] for k_ in ("scorer_calls", "poses_evaluated", "launches")})
