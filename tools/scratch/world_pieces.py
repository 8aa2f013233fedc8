"""scratch: where a scan of bench.py's world_loop leg goes (wall time of each C-ABI call through the Python layer)"""
import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import __graft_entry__ as ge
from synth import make_scene
pkg = ge.load_package()
sc = make_scene(cell_model=0, size=2000, scale=0.05, n_beams=1080, seed=100, weighting="even")
scan, m0 = sc["scan"], sc["map"]
ctx = pkg.Context(0)
cos_a, sin_a = pkg.beam_trig(scan.angle)
ctx.map_bind(5, m0.cell_model, m0.width, m0.height, m0.origin, m0.scale, m0.unknown)
ctx.map_upload_window(5, 0, 0, m0.payload)
ctx.map_set_auto_grow(5, True)
m = pkg.Matcher(ctx, "HC", pkg.spe_cfg(), [128, 0.1, 0.1])
init = np.asarray(sc["init_pose"], dtype=np.float64)
for deferred in (True, False):
    ctx.map_set_deferred(deferred)
    t = np.zeros(3)
    K = 100
    for k in range(K + 5):
        a = time.perf_counter()
        ctx.scan_upload(scan.range, cos_a, sin_a, scan.weight, scan.factor)
        b = time.perf_counter()
        r = m.process_scan(5, init)
        c = time.perf_counter()
        ctx.map_append_scan(5, pkg.RULE_MEAN, init + r["delta"], scan.range, cos_a, sin_a)
        d = time.perf_counter()
        if k >= 5:
            t += [b - a, c - b, d - c]
    ctx.map_drain() if deferred else None
    print("deferred %s: scan_upload %.1f us, process_scan %.1f us, map_append_scan %.1f us" % ((deferred,) + tuple(1e6 * t / K)))
