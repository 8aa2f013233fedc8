"""Per-scene counters of the cfg2 match at SLAMHIP_OPT_INERT_TAIL = 0 / 1 (diagnostic)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402
from bench_legs.common import WORKLOADS, rotating_scenes  # noqa: E402
from synth import make_scene  # noqa: E402

pkg = ge.load_package()
cell, weighting, kind, params, _, _ = WORKLOADS["hc"]
sc = make_scene(cell_model=cell, size=2000, scale=0.05, n_beams=1080, seed=100, weighting=weighting)
scenes = rotating_scenes(sc, 1080, weighting)
ctx = pkg.Context(0)
ctx.upload_map(0, sc["map"])
for j, s_ in enumerate(scenes):
    c_, s__ = pkg.beam_trig(s_["angle"])
    ctx.scan_store(j, s_["range"], c_, s__, s_["weight"])
tot = {}
for level in (0, 1):
    ctx.set_option(pkg.OPT_INERT_TAIL, level)
    m = pkg.Matcher(ctx, kind, pkg.spe_cfg(), params)
    rows = []
    for k in range(len(scenes)):
        ctx.scan_select(k)
        r = m.process_scan(0, scenes[k]["init_pose"])
        st = m.stats()
        rows.append((st["scorer_calls"], st["poses_evaluated"], st["launches"], st["calls_closed_form"], r["prob"]))
    tot[level] = rows
    m.close()
for k in range(len(scenes)):
    print(k, " | ".join("calls %d eval %d steps %d closed %d" % tot[l][k][:4] for l in (0, 1)),
          "same" if tot[0][k][4] == tot[1][k][4] else "DIFFERENT")
for l in (0, 1):
    print("level", l, "mean steps %.2f" % np.mean([r[2] for r in tot[l]]))
