#!/usr/bin/env python3
"""G5b: GmappingParticleFilter steps WITH the map update (the reference's default: every particle
appends its scan to the ONE shared map before the next particle matches, Q20) captured from the
compiled reference -> tests/golden/gmapping_pf_update.npz: per step poses / weights / master flags /
resampling decision and the shared map's payload + (hits, tries) after the step."""
import os
import sys

import numpy as np

GOLDEN_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(GOLDEN_DIR))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as po  # noqa: E402
from pyoracle import RefGmapping  # noqa: E402


def main():
    R = po.Ref()
    scale, n_cells, n = 0.05, 512, 8
    gt = R.map_create(po.REF_CELL_MOCK, po.MAP_UNBOUNDED_PLAIN, n_cells, n_cells, scale, 0.0)
    gt.stamp_text(R.cecum_text(61, 45, 2), (-30, 20), 2, 2)
    gt.stamp_text(R.cecum_text(25, 17, 3), (-12, -8), 2, 2)
    pose0 = np.array([scale / 2, scale / 2 - 6 * scale, np.deg2rad(90)])
    gp = [0.0, 0.05, 0.0, 0.02, 0.0, 0.0, 0.0, 0.0]
    seeds = np.arange(2000, 2000 + n, dtype=np.uint32)
    g = RefGmapping(R, n, n_cells, n_cells, scale, gp, seeds, skip_rate=3, blur=0.0, map_max_range=np.inf)
    mview = g.map()
    g0 = mview.geometry()
    deltas = [pose0, [0.1, 0.05, 0.04], [0.15, -0.05, 0.06], [0.1, 0.1, -0.05], [0.2, 0.0, 0.08]]
    out = dict(gp=np.array(gp), seeds=seeds, scale=np.array(scale), n_steps=np.array(len(deltas)),
               origin=np.array(g0["origin"]), size=np.array([g0["width"], g0["height"]]),
               unknown=mview.to_data().unknown)
    true = np.zeros(3)
    for k, d in enumerate(deltas):
        true = true + np.asarray(d)
        tp = true.copy()
        tp[:2] = (np.floor(tp[:2] / scale) + 0.5) * scale
        scan = R.scan_generate(gt, tp, 8, 270, 360)
        r, a, o, _ = scan.get()
        res, poses, w, ms = g.step(scan, d, 7 + k, np.arange(9000 + 100 * k, 9000 + 100 * k + n, dtype=np.uint32))
        assert mview.geometry() == g0, "the map must not grow in this fixture"
        out["step%d_range" % k], out["step%d_angle" % k] = r, a
        out["step%d_delta" % k] = np.asarray(d, dtype=np.float64)
        out["step%d_resampled" % k] = np.array(int(res))
        out["step%d_poses" % k], out["step%d_weights" % k], out["step%d_master" % k] = poses, w, ms
        out["step%d_payload" % k] = mview.to_data().payload.copy()
        out["step%d_aux" % k] = mview.aux().copy()
    path = os.path.join(GOLDEN_DIR, "gmapping_pf_update.npz")
    np.savez_compressed(path, **out)
    print("wrote gmapping_pf_update.npz", os.path.getsize(path) // 1024, "KiB",
          [int(out["step%d_resampled" % k]) for k in range(len(deltas))])


if __name__ == "__main__":
    main()
