#!/usr/bin/env python3
"""G6b: map update with the AreaOccupancyEstimator (slam/occupancy_estimator/type = area) captured
from the compiled reference -> tests/golden/map_update_area.npz.

Q27: ensure_segment_not_on_edge freezes Shift_Amount = low_qual * cell.side() in a function-local
static at the estimator's FIRST call in the process.  This script therefore runs in a fresh process
and first inserts one beam that ends in the cell (0, 0), whose side is scale*(0+1) - scale*0 =
scale exactly, so the static is 0.01 * scale -- the value the tests pass to the restatements."""
import os
import sys

import numpy as np

GOLDEN_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(GOLDEN_DIR))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as po  # noqa: E402


def main():
    R = po.Ref()
    scale, n = 0.1, 340
    # pin the static (see the module docstring): beam from the middle of cell (0,0), length 0.01
    pin = R.map_create(po.REF_CELL_MEAN, po.MAP_UNBOUNDED_PLAIN, 20, 20, scale)
    R.append_scan(pin, R.scan_create([0.01], [0.0]), (0.05, 0.05, 0.0), occ_est=1)
    gt = R.map_create(po.REF_CELL_MOCK, po.MAP_UNBOUNDED_PLAIN, n, n, scale, 0.0)
    gt.stamp_text(R.cecum_text(61, 45, 2), (-30, 20))
    gt.stamp_text(R.cecum_text(25, 17, 3), (-12, -8))
    poses = [(0.05, -0.25, np.deg2rad(90)), (0.45, 0.35, np.deg2rad(60)), (-0.52, -0.13, np.deg2rad(125)),
             (0.0, 0.0, 0.0)]  # the last pose sits on cell edges with axis-aligned beams (edge shift)
    steps = [dict(quality=1.0, blur=0.0, max_range=np.inf), dict(quality=0.9, blur=0.3, max_range=np.inf),
             dict(quality=0.7, blur=0.1, max_range=8.0), dict(quality=1.0, blur=0.0, max_range=np.inf)]
    out = dict(scale=np.array(scale), n_steps=np.array(len(steps)), shift_amount=np.array(0.01 * scale))
    rs = np.random.RandomState(12)
    scans = []
    for k, p in enumerate(poses):
        if k < 3:
            sc = R.scan_generate(gt, p, 15, 270, 360)
            r, a, o, _ = sc.get()
            o = o.copy()
            miss = rs.rand(r.size) < 0.08
            o[miss] = 0
            r = np.where(miss, 12.0, r + rs.randn(r.size) * 0.005)
        else:
            a = np.deg2rad(np.array([0.0, 90.0, 180.0, -90.0, 45.0, 135.0, 30.0]))
            r = np.array([1.0, 0.75, 1.25, 0.5, np.sqrt(2.0), 2 * np.sqrt(2.0), 1.3])
            o = np.ones(a.size, np.int32)
        scans.append((r, a, o))
        out["step%d_pose" % k] = np.array(p)
        out["step%d_range" % k], out["step%d_angle" % k], out["step%d_occ" % k] = r, a, o
        out["step%d_params" % k] = np.array([steps[k]["quality"], steps[k]["blur"], steps[k]["max_range"]])
    models = {"mean": (po.REF_CELL_MEAN, (0.95, 1.0, 0.01, 1.0)), "tbm": (po.REF_CELL_TBM, (0.95, 0.04, 0.01, 0.003)),
              "gmapping": (po.REF_CELL_GMAPPING, (0.95, 1.0, 0.01, 1.0))}
    for name, (cell, base) in models.items():
        mtype = po.MAP_UNBOUNDED_LAZY_TILED if cell == po.REF_CELL_GMAPPING else po.MAP_UNBOUNDED_PLAIN
        m = R.map_create(cell, mtype, n, n, scale, 0.5)
        g0 = m.geometry()
        out[name + "_base"] = np.array(base)
        out[name + "_origin"] = np.array(g0["origin"])
        out[name + "_size"] = np.array([g0["width"], g0["height"]])
        out[name + "_unknown"] = m.to_data().unknown
        for k, p in enumerate(poses):
            r, a, o = scans[k]
            R.append_scan(m, R.scan_create(r, a, o), p, quality=steps[k]["quality"], occ_est=1, base=base,
                          blur=steps[k]["blur"], max_range=steps[k]["max_range"])
            assert m.geometry() == g0
            md = m.to_data()
            out["%s_step%d_payload" % (name, k)] = md.payload[30:310, 30:310].copy()
            aux = m.aux()
            if aux is not None:
                out["%s_step%d_aux" % (name, k)] = aux[30:310, 30:310].copy()
    out["crop"] = np.array([30, 310])
    path = os.path.join(GOLDEN_DIR, "map_update_area.npz")
    np.savez_compressed(path, **out)
    print("wrote map_update_area.npz", os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
