"""In-kernel timeline of the hill-climbing chain (csrc/hc_chain.hip): wall-clock stamps of one scoring
workgroup per super-step -- where a super-step's microseconds go.  Run on the GPU box."""
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
ctx = pkg.Context(0, testing=True)  # (the stamps are a hook of libslamhip_testing.so)
sc = make_scene(cell_model=0, size=2000, scale=0.05, n_beams=1080, seed=100)
ctx.upload_map(0, sc["map"])
c, s = pkg.beam_trig(sc["scan"].angle)
ctx.scan_upload(sc["scan"].range, c, s, sc["scan"].weight, sc["scan"].factor)
cases = [(256, 1, False), (512, 1, False), (512, 0, False), (1024, 1, False), (1024, 1, True)]
for threads, check, gm in cases:  # check 0: without the tie check; gm: the GMapping OOPE on a GMapping map
    if gm:
        from synth import CELL_GMAPPING
        sc = make_scene(cell_model=CELL_GMAPPING, size=2000, scale=0.05, n_beams=1080, seed=100)
        ctx.upload_map(0, sc["map"])
        c, s = pkg.beam_trig(sc["scan"].angle)
        ctx.scan_upload(sc["scan"].range, c, s, sc["scan"].weight, sc["scan"].factor)
    m = pkg.Matcher(ctx, "HC", pkg.spe_cfg(oope=pkg.OOPE_GMAPPING) if gm else pkg.spe_cfg(), [20 if gm else 128, 0.1, 0.1])
    m.set_device_chain(1, threads)
    m.set_tie_check(check)
    for _ in range(5):
        m.process_scan(0, sc["init_pose"])
    L = pkg.load(testing=True)
    L.slamhip_matcher_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
    L.slamhip_matcher_debug_stamps(m.h, None)
    m.process_scan(0, sc["init_pose"])
    buf = (C.c_longlong * 512)()
    L.slamhip_matcher_debug_stamps(m.h, buf)
    st = np.array(list(buf)).reshape(64, 8)
    steps = m.stats()["launches"]
    st = st[:steps]
    print("threads %d, tie check %d, gmapping %d: %d super-steps, %d re-scored" % (threads, check, gm, steps, m.stats()["steps_rescored"]))
    names = ["staged", "replayed", "pose", "terms (GMapping: phase A)", "stored (GMapping: run cache + sum)"]
    d = np.diff(st[:, :6], axis=1) / 100.0
    ok = (st[:, 5] > 0)
    print("  us per phase (mean over the super-steps that scored): " +
          ", ".join("%s %.2f" % (n, v) for n, v in zip(names, d[ok].mean(0))))
    print("  inside the replay: records in registers +%.2f, outcomes / ballots +%.2f, advance + broadcast +%.2f us" %
          ((st[ok, 6] - st[ok, 1])[1:].mean() / 100.0, (st[ok, 7] - st[ok, 6])[1:].mean() / 100.0,
           (st[ok, 2] - st[ok, 7])[1:].mean() / 100.0))
    print("  entry -> stored: %.2f us; entry(k+1) - entry(k): %.2f us; stored(k) -> entry(k+1): %.2f us" %
          ((st[ok, 5] - st[ok, 0]).mean() / 100.0, np.diff(st[:, 0]).mean() / 100.0,
           (st[1:, 0] - st[:-1, 5])[ok[:-1]].mean() / 100.0))
