"""In-kernel timeline of chain 0 of a co-resident BATCH (slamhip_matcher_process_scan_batch, csrc/hc_resident.hip): where a
super-step's microseconds go when K chains share the chip.  Run on the GPU box (uses libslamhip_testing.so: the stamps are
a testing hook)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402
from synth import make_scene  # noqa: E402

pkg = ge.load_package()
ctx = pkg.Context(0, testing=True)
L = pkg.load(testing=True)
L.slamhip_matcher_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
sc = make_scene(cell_model=0, size=2000, scale=0.05, n_beams=1080, seed=100)
ctx.upload_map(0, sc["map"])
c, s = pkg.beam_trig(sc["scan"].angle)
rs = np.random.RandomState(3)
for k in range(16):
    ctx.scan_store(k, sc["scan"].range, c, s, sc["scan"].weight)
for K in (int(x) for x in (sys.argv[1:] or ["2", "4", "8", "16"])):
    m = pkg.Matcher(ctx, "HC", pkg.spe_cfg(), [128, 0.1, 0.1])
    jobs = m.make_batch([dict(map_id=0, scan_slot=k, init_pose=sc["true_pose"] + rs.randn(3) * [0.07, 0.05, 0.03]) for k in range(K)])
    for _ in range(4):
        m.process_scan_batch(jobs)
    L.slamhip_matcher_debug_stamps(m.h, None)
    m.process_scan_batch(jobs)
    buf = (C.c_longlong * 512)()
    L.slamhip_matcher_debug_stamps(m.h, buf)
    st = np.array(list(buf)).reshape(64, 8)
    steps = min(m.batch_stats(0)["super_steps"], 64)
    st = st[:steps]
    ok = st[:, 5] > 0
    us = lambda a, b: ((st[ok, a] - st[ok, b]) / 100.0).mean()  # noqa: E731
    print("K = %d: chain 0 took %d super-steps (%d scored by workgroup 1); resident %r" % (K, steps, ok.sum(), m.resident_stats()))
    # (a pair's stamping thread -- slot 1 = the workgroup's second pose -- does not sweep: the replay's stamps stay empty)
    swept = st[ok, 1].min() > 0
    print("  us per phase: pose %.2f, terms %.2f, sum + publish %.2f%s; super-step to super-step %.2f"
          % (us(3, 0), us(4, 3), us(5, 4),
             (", publish -> all scores here %.2f, decisions %.2f, rest of replay %.2f" % (us(1, 5), us(7, 1), us(2, 7))) if swept else "",
             (np.diff(st[:, 0]) / 100.0).mean()))
    m.close()
