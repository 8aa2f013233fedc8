#!/usr/bin/env python3
"""tools/k6_single_probe.py [rule] [n_scans] -- single-scan map updates (slamhip_map_append_scan, counting-sorted K6)
on the headline's scene for profiling:  rocprofv3 --kernel-trace --stats -- python3 tools/k6_single_probe.py
rule: 2 MeanProbabilityCell on an OCC map (the world loop's), 4 GmappingBaseCell (the filter's shared-map step)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import __graft_entry__ as ge  # noqa: E402
from synth import make_scene  # noqa: E402

pkg = ge.load_package()
rule = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n_scans = int(sys.argv[2]) if len(sys.argv) > 2 else 200
size = 2000
sc = make_scene(cell_model=2 if rule == 4 else 0, size=size, scale=0.05, n_beams=1080, seed=100)
ctx = pkg.Context(0)
ctx.upload_map(1, sc["map"])
scan = sc["scan"]
c, s = pkg.beam_trig(scan.angle)
rs = np.random.RandomState(5)
poses = sc["true_pose"] + rs.randn(n_scans + 8, 3) * [0.05, 0.05, 0.02]
for k in range(8):
    nu = ctx.map_append_scan(1, rule, poses[k], scan.range, c, s)
ctx.synchronize()
t0 = time.perf_counter()
for k in range(n_scans):
    ctx.map_append_scan(1, rule, poses[8 + k], scan.range, c, s)
ctx.synchronize()
dt = time.perf_counter() - t0
print("awaited: %.1f us per update, %d cell updates each" % (1e6 * dt / n_scans, nu))
ctx.map_set_deferred(True)
t0 = time.perf_counter()
for k in range(n_scans):
    ctx.map_append_scan(1, rule, poses[8 + k], scan.range, c, s)
ctx.map_drain()
dt = time.perf_counter() - t0
print("queued:  %.1f us per update" % (1e6 * dt / n_scans))
