"""Soak of the closed-form tail in batches (diagnostic): random K, random initial poses, SLAMHIP_OPT_INERT_TAIL 2 vs 0."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402
from bench_legs.common import rotating_scenes  # noqa: E402
from synth import make_scene  # noqa: E402

pkg = ge.load_package()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rs = np.random.RandomState(7)
bad = jobs_done = closed = nonres = 0
t0 = time.time()
for cell, weighting in ((0, "even"), (1, "viny")):
    sc = make_scene(cell_model=cell, size=2000, scale=0.05, n_beams=1080, seed=100 + cell, weighting=weighting)
    scenes = rotating_scenes(sc, 1080, weighting)
    ctx = pkg.Context(0)
    ctx.upload_map(0, sc["map"])
    for j, s_ in enumerate(scenes):
        c_, s__ = pkg.beam_trig(s_["angle"])
        ctx.scan_store(j, s_["range"], c_, s__, s_["weight"])
    for prm in ([128, 0.1, 0.1], [80, 0.05, 0.2]):
        ms = {}
        for level in (2, 0):
            ctx.set_option(pkg.OPT_INERT_TAIL, level)
            ms[level] = pkg.Matcher(ctx, "HC", pkg.spe_cfg(), prm)
            ms[level].set_device_chain(2)
        for i in range(N // 4):
            K = int(rs.choice([2, 3, 5, 8, 9, 10, 13, 16, 24, 28, 32, 48, 64]))
            jobs = []
            for _ in range(K):
                k = int(rs.randint(16))
                jobs.append(dict(map_id=0, scan_slot=k, init_pose=scenes[k]["true_pose"] + rs.randn(3) * [0.1, 0.1, 0.05]))
            out = {}
            for level in (2, 0):
                ctx.set_option(pkg.OPT_INERT_TAIL, level)
                r = ms[level].process_scan_batch(jobs)
                st = [ms[level].batch_stats(j)["scorer_calls"] for j in range(K)]
                out[level] = [(x["prob"], tuple(x["delta"]), c) for x, c in zip(r, st)]
                if level == 2:
                    closed += ms[level].stats()["calls_closed_form"] > 0
            jobs_done += K
            if out[2] != out[0]:
                bad += 1
                print("DIFFERENT", cell, prm, K)
        for level in (2, 0):
            nonres += ms[level].resident_stats()["gave_up"]
            ms[level].close()
    ctx.set_option(pkg.OPT_INERT_TAIL, 2)
    ctx.close()
print("%d batches (%d matches), %d with closed-form tails, %d different, %d give-ups, %.0f s" % (N, jobs_done, closed, bad, nonres, time.time() - t0))
