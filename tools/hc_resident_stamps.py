"""In-kernel timeline of the co-resident hill-climbing chain (csrc/hc_resident.hip): wall-clock stamps of one scoring
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
L = pkg.load(testing=True)
L.slamhip_matcher_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
for threads, check in [(1024, 1), (1024, 0), (512, 1), (256, 1), (1024, -1), (512, 2)]:
    mc = check == 2  # the Monte-Carlo matcher (csrc/mc_resident.hip), BASELINE configs[2]'s parameters on a TBM map
    if mc:
        sc = make_scene(cell_model=1, size=2000, scale=0.05, n_beams=1080, seed=100, weighting="viny")
        ctx.upload_map(0, sc["map"])
        c, s = pkg.beam_trig(sc["scan"].angle)
        ctx.scan_upload(sc["scan"].range, c, s, sc["scan"].weight, sc["scan"].factor)
        check = 1
    gm = check < 0  # the GMapping OOPE on a GMapping map (csrc/hc_resident_gm.hip)
    if gm:
        from synth import CELL_GMAPPING
        sc = make_scene(cell_model=CELL_GMAPPING, size=2000, scale=0.05, n_beams=1080, seed=100)
        ctx.upload_map(0, sc["map"])
        c, s = pkg.beam_trig(sc["scan"].angle)
        ctx.scan_upload(sc["scan"].range, c, s, sc["scan"].weight, sc["scan"].factor)
        check = 0
    if mc:
        m = pkg.Matcher(ctx, "MC", pkg.spe_cfg(), [666666, 0.2, 0.1, 4096, 4096])
    else:
        m = pkg.Matcher(ctx, "HC", pkg.spe_cfg(oope=pkg.OOPE_GMAPPING) if gm else pkg.spe_cfg(), [20 if gm else 128, 0.1, 0.1])
    m.set_device_chain(2, threads)
    m.set_tie_check(check)
    for _ in range(5):
        m.process_scan(0, sc["init_pose"])
    L.slamhip_matcher_debug_stamps(m.h, None)
    m.process_scan(0, sc["init_pose"])
    buf = (C.c_longlong * 512)()
    L.slamhip_matcher_debug_stamps(m.h, buf)
    st = np.array(list(buf)).reshape(64, 8)
    steps = min(m.stats()["launches"], 64)
    st = st[:steps]
    ok = st[:, 5] > 0  # super-steps in which workgroup 1 scored a pose
    us = lambda a, b: ((st[ok, a] - st[ok, b]) / 100.0).mean()
    print("%sthreads %d, tie check %d: %d super-steps (%d scored by workgroup 1), %d re-scored, resident %r" %
          ("Monte Carlo, " if mc else ("GMapping OOPE, " if gm else ""), threads, check, steps, ok.sum(), m.stats()["steps_rescored"], m.resident_stats()))
    print("  us per phase: pose %.2f, terms (GMapping: phase A) %.2f, sum + publish (GMapping: run cache + sum + publish) %.2f, publish -> all scores here %.2f, "
          "decisions + ballots %.2f, advance %.2f" % (us(3, 0), us(4, 3), us(5, 4), us(1, 5), us(7, 1), us(2, 7)))
    nxt = (st[1:, 0] - st[:-1, 2]) / 100.0
    print("  replay end -> next super-step's start %.2f; super-step to super-step %.2f us" %
          (nxt.mean(), (np.diff(st[:, 0]) / 100.0).mean()))
