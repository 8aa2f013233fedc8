"""Soak of the closed-form tail (diagnostic): N random matches over the bench's scenes -- random initial poses, limits and
step sizes -- at SLAMHIP_OPT_INERT_TAIL 2 and 0; result (pose delta, probability) and scorer-call count must be equal."""
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
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rs = np.random.RandomState(99)
bad = closed = 0
t0 = time.time()
done = 0
for cell, weighting, scale in ((0, "even", 0.05), (1, "viny", 0.05), (0, "even", 0.2)):
    sc = make_scene(cell_model=cell, size=2000 if scale == 0.05 else 500, scale=scale, n_beams=1080, seed=100 + cell, weighting=weighting)
    scenes = rotating_scenes(sc, 1080, weighting)
    ctx = pkg.Context(0)
    ctx.upload_map(0, sc["map"])
    for j, s_ in enumerate(scenes):
        c_, s__ = pkg.beam_trig(s_["angle"])
        ctx.scan_store(j, s_["range"], c_, s__, s_["weight"])
    for prm in ([128, 0.1, 0.1], [60, 0.05, 0.2], [200, 0.3, 0.02], [128, 0.01, 0.01]):
        ms = {}
        for level in (2, 0):
            ctx.set_option(pkg.OPT_INERT_TAIL, level)
            ms[level] = pkg.Matcher(ctx, "HC", pkg.spe_cfg(), prm)
        ctx.set_option(pkg.OPT_INERT_TAIL, 2)
        for i in range(N // 12):
            k = int(rs.randint(16))
            pose = scenes[k]["true_pose"] + rs.randn(3) * [0.1, 0.1, 0.05]
            ctx.scan_select(k)
            out = {}
            for level in (2, 0):
                ctx.set_option(pkg.OPT_INERT_TAIL, level)
                r = ms[level].process_scan(0, pose)
                st = ms[level].stats()
                out[level] = (r["prob"], tuple(r["delta"]), st["scorer_calls"])
                if level == 2:
                    closed += st["calls_closed_form"] > 0
            done += 1
            if out[2] != out[0]:
                bad += 1
                print("DIFFERENT", cell, prm, k, pose, out)
        for m in ms.values():
            m.close()
    ctx.set_option(pkg.OPT_INERT_TAIL, 2)
    ctx.close()
print("%d matches, %d with a closed-form tail, %d different, %.0f s" % (done, closed, bad, time.time() - t0))
