"""Per-job super-steps of a K-match batch of the bench scenes at SLAMHIP_OPT_INERT_TAIL 0 / 1 / 2 (diagnostic)."""
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
K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rows = {}
for level in (0, 1, 2):
    ctx.set_option(pkg.OPT_INERT_TAIL, level)
    m = pkg.Matcher(ctx, kind, pkg.spe_cfg(), params)
    m.set_device_chain(2)
    blk = m.make_batch([dict(map_id=0, scan_slot=k % 16, init_pose=scenes[k % 16]["init_pose"]) for k in range(K)])
    m.process_scan_batch(blk)
    r = m.process_scan_batch(blk)
    rows[level] = ([m.batch_stats(j) for j in range(K)], m.stats(), m.resident_stats(), [x["prob"] for x in r])
    m.close()
for j in range(min(K, 16)):
    print(j, " | ".join("steps %d eval %d" % (rows[l][0][j]["super_steps"], rows[l][0][j]["poses_evaluated"]) for l in (0, 1, 2)),
          "same" if rows[0][3][j] == rows[1][3][j] == rows[2][3][j] else "DIFFERENT")
for l in (0, 1, 2):
    print("level", l, "longest", rows[l][1]["launches"], "closed", rows[l][1]["calls_closed_form"], rows[l][2])
