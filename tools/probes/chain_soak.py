"""Soak of the device chains: many matches in a row, every result compared (HC: with the first one; MC: with a
host-driven twin that draws the same random stream; the filter: chains against a lock-step twin is a pytest)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import __graft_entry__ as ge
from synth import make_scene
pkg = ge.load_package()
ctx = pkg.Context(0)
sc = make_scene(cell_model=0, size=2000, scale=0.05, n_beams=1080, seed=100)
ctx.upload_map(0, sc["map"])
c, s = pkg.beam_trig(sc["scan"].angle)
ctx.scan_upload(sc["scan"].range, c, s, sc["scan"].weight, sc["scan"].factor)
m = pkg.Matcher(ctx, "HC", pkg.spe_cfg(), [128, 0.1, 0.1])
first = m.process_scan(0, sc["init_pose"])
t0 = time.time()
n_hc = int(os.environ.get("SOAK_HC", "20000"))
for k in range(n_hc):
    r = m.process_scan(0, sc["init_pose"])
    assert r["prob"] == first["prob"] and np.array_equal(r["delta"], first["delta"]), k
print("HC chain: %d matches identical, %.1f s; co-resident launches %r" % (n_hc, time.time() - t0, m.resident_stats()))
dev = pkg.Matcher(ctx, "MC", pkg.spe_cfg(), [99, 0.2, 0.1, 200, 600])
host = pkg.Matcher(ctx, "MC", pkg.spe_cfg(), [99, 0.2, 0.1, 200, 600])
host.set_device_chain(0)
t0 = time.time()
n_mc = int(os.environ.get("SOAK_MC", "3000"))
init = np.array(sc["init_pose"])
for k in range(n_mc):
    a, b = dev.process_scan(0, init), host.process_scan(0, init)
    assert a["prob"] == b["prob"] and np.array_equal(a["delta"], b["delta"]), k
    init = sc["init_pose"] + 0.02 * np.array([np.sin(k), np.cos(1.3 * k), 0.3 * np.sin(0.7 * k)])
print("MC chain: %d matches equal to the host-driven twin, %.1f s; co-resident launches %r" % (n_mc, time.time() - t0, dev.resident_stats()))
