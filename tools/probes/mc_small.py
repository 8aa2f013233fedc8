"""cfg1-like Monte Carlo (100 attempts, 20 failures): device chain against host-driven batches, ms per match"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import __graft_entry__ as ge
from synth import make_scene
pkg = ge.load_package()
ctx = pkg.Context(0)
sc = make_scene(cell_model=0, size=1000, scale=0.1, n_beams=720, seed=1)
ctx.upload_map(0, sc["map"])
c, s = pkg.beam_trig(sc["scan"].angle)
ctx.scan_upload(sc["scan"].range, c, s, sc["scan"].weight, sc["scan"].factor)
for prm in ([666666, 0.2, 0.1, 20, 100], [666666, 0.2, 0.1, 100, 400], [666666, 0.2, 0.1, 4096, 4096]):
    for mode in (1, 0):
        m = pkg.Matcher(ctx, "MC", pkg.spe_cfg(), prm)
        m.set_device_chain(mode)
        for _ in range(5):
            m.process_scan(0, sc["init_pose"])
        t0 = time.perf_counter()
        n = 50
        for _ in range(n):
            m.process_scan(0, sc["init_pose"])
        dt = (time.perf_counter() - t0) / n
        st = m.stats()
        print(prm[3:], "chain" if mode else "host ", "%.3f ms" % (dt * 1e3), st["scorer_calls"], st["launches"])
