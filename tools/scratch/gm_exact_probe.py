"""Probe (r06): the exact GMapping scorer against the oracle on a synthetic scene -- which ingredient differs."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import __graft_entry__ as ge  # noqa: E402
import pyoracle as po  # noqa: E402
from synth import CELL_GMAPPING, make_scene  # noqa: E402

pkg = ge.load_package()
print("libm variant", pkg.libm_variant())
ctx = pkg.Context(0)
O = po.Oracle()
sc = make_scene(cell_model=CELL_GMAPPING, size=500, scale=0.05, n_beams=360, seed=540)
ctx.upload_map(0, sc["map"])
s = sc["scan"]
c, sn = pkg.beam_trig(s.angle)
ctx.scan_upload(s.range, c, sn, s.weight, s.factor)
ctx.scan_set_angles(s.angle)
rs = np.random.RandomState(1)
poses = sc["true_pose"] + rs.randn(64, 3) * [0.08, 0.08, 0.04]
for name, kw in (("seq+raw_exact", dict(sum_order=1, pose_trig=2)), ("seq+host", dict(sum_order=1, pose_trig=1))):
    ctx.gm_cache_reset()
    got = ctx.score_poses(0, pkg.spe_cfg(oope=pkg.OOPE_GMAPPING, **kw), poses)
    want = O.score_poses(sc["map"], s, po.make_cfg(oope=po.OOPE_GMAPPING), poses, po.Oracle.new_gm_cache())
    tr = po.ScanData(s.range, np.arange(s.n, dtype=np.float64), s.weight, s.factor, po.TRIG_CACHED, 0.0, 1.0, sn, c)
    want_c = O.score_poses(sc["map"], tr, po.make_cfg(oope=po.OOPE_GMAPPING), poses, po.Oracle.new_gm_cache())
    print(name, "vs oracle raw: equal %d/64 max rel %.2e | vs oracle cached-arith: equal %d/64 max rel %.2e"
          % ((got == want).sum(), np.max(np.abs(got / want - 1)), (got == want_c).sum(), np.max(np.abs(got / want_c - 1))))
print("weights", s.weight[:3], "factor", s.factor[:3], "unknown", sc["map"].unknown)
