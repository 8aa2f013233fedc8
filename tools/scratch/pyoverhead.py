"""scratch: how much of a headline step is the Python wrapper (Matcher.process_scan + stats) and how much the C-ABI call"""
import sys, time, ctypes as C
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import __graft_entry__ as ge
from synth import make_scene
pkg = ge.load_package()
sc = make_scene(cell_model=0, size=2000, scale=0.05, n_beams=1080, seed=100, weighting="even")
scan = sc["scan"]
ctx = pkg.Context(0)
ctx.upload_map(0, sc["map"])
c, s = pkg.beam_trig(scan.angle)
ctx.scan_upload(scan.range, c, s, scan.weight, scan.factor)
m = pkg.Matcher(ctx, "HC", pkg.spe_cfg(), [128, 0.1, 0.1])
K = 300
for _ in range(10):
    m.process_scan(0, sc["init_pose"])
t0 = time.perf_counter()
for _ in range(K):
    m.process_scan(0, sc["init_pose"]); m.stats()["scorer_calls"]
t1 = time.perf_counter()
L = m.L
ip = np.ascontiguousarray(sc["init_pose"], dtype=np.float64)
delta = np.zeros(3); prob = C.c_double()
dp = C.POINTER(C.c_double)
a_ip, a_d = ip.ctypes.data_as(dp), delta.ctypes.data_as(dp)
for _ in range(K):
    L.slamhip_matcher_process_scan(m.h, 0, a_ip, a_d, C.byref(prob))
t2 = time.perf_counter()
print("wrapper %.2f us/step, raw C-ABI call %.2f us/step" % (1e6 * (t1 - t0) / K, 1e6 * (t2 - t1) / K))
print({k: round(v, 1) if isinstance(v, float) else v for k, v in m.stats().items()})
