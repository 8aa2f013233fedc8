import ctypes as C, os, sys
import numpy as np
ROOT = "/root/repo" if os.path.isdir("/root/repo") else os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
from synth import make_scene, CELL_GMAPPING
pkg = ge.load_package()
ctx = pkg.Context(0, testing=True)
L = pkg.load(testing=True)
L.slamhip_matcher_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
sc = make_scene(cell_model=CELL_GMAPPING, size=2000, scale=0.05, n_beams=1080, seed=100)
ctx.upload_map(0, sc["map"])
c, s = pkg.beam_trig(sc["scan"].angle)
ctx.scan_upload(sc["scan"].range, c, s, sc["scan"].weight, sc["scan"].factor)
m = pkg.Matcher(ctx, "HC", pkg.spe_cfg(oope=pkg.OOPE_GMAPPING), [20, 0.1, 0.1])
m.set_device_chain(2, 1024); m.set_tie_check(0)
for _ in range(5): m.process_scan(0, sc["init_pose"])
L.slamhip_matcher_debug_stamps(m.h, None)
m.process_scan(0, sc["init_pose"])
buf = (C.c_longlong * 512)(); L.slamhip_matcher_debug_stamps(m.h, buf)
a = np.array(list(buf))
st = a[:256].reshape(32, 8); fine = a[256:].reshape(32, 8)
n = min(m.stats()["launches"], 15)
ok = [k for k in range(1, n) if st[k, 5] > 0 and fine[k, 4] > 0]
f = lambda x, y: np.mean([(x[k] - y[k]) / 100.0 for k in ok])
print("steps", len(ok))
print("phaseA end -> b2 %.2f, b2 -> b3 %.2f, b3 -> sum loop done %.2f, xor sum %.2f, last barrier %.2f, tail->publish %.2f" % (
    f(fine[:, 0], st[:, 4]), f(fine[:, 1], fine[:, 0]), f(fine[:, 2], fine[:, 1]), f(fine[:, 3], fine[:, 2]), f(fine[:, 4], fine[:, 3]), f(st[:, 5], fine[:, 4])))
print("pose %.2f phaseA %.2f" % (f(st[:, 3], st[:, 0]), f(st[:, 4], st[:, 3])))
