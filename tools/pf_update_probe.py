#!/usr/bin/env python3
"""tools/pf_update_probe.py -- a few GMapping steps WITH the shared-map update (100 particles,
1080 beams, 4000x4000 @0.05 m) for profiling:  rocprofv3 --kernel-trace --stats -- python3 tools/pf_update_probe.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import __graft_entry__ as ge  # noqa: E402
from synth import make_scene  # noqa: E402

pkg = ge.load_package()
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
sc = make_scene(cell_model=2, size=size, scale=0.05, n_beams=1080, seed=4)
ctx = pkg.Context(0)
ctx.upload_map(1, sc["map"])
n = 100
pf = pkg.GmappingFilter(ctx, pkg.gmapping_params(gp8=[0.0, 0.1, 0.0, 0.03, 0, 0, 0, 0]), n,
                        np.arange(1000, 1000 + n, dtype=np.uint32))
pf.set_map_update(True)
scan = sc["scan"]
rs = np.random.RandomState(5)
pf.step(1, scan.range, scan.angle, None, sc["true_pose"], 7)
t0 = time.perf_counter()
for k in range(3):
    pf.step(1, scan.range, scan.angle, None, rs.randn(3) * [0.05, 0.05, 0.02], 8 + k)
dt = time.perf_counter() - t0
print("ms per step", 1e3 * dt / 3, pf.stats())
