#!/usr/bin/env python3
"""G6b: the map update with a PER-POINT observation quality -> tests/golden/map_update_ahr.npz.

The reference's scan adder multiplies the scan quality by `_omqe->quality(points, pt_i)`
(src/core/maps/grid_map_scan_adders.h:54-75); init_omqe (src/utils/init_occupancy_mapping.h:64-80) builds IdleOMQE
(1.0) or, for `slam/mapping/observation_quality_estimator/typetype = ahr` (the reference's own misspelt key),
AngleHistogramResiprocalOMQE (grid_map_scan_adders.h:32-43: 1 / AngleHistogram::value).  Captured from the compiled
reference: the per-point qualities of three scans, and -- for the three cell kinds whose `+=` reads the observation's
quality (AffineQualityMergeCell, MeanProbabilityCell, TbmBaseCell: naive_grid_cells.h:14-40, tbm_grid_cells.h:57-66)
and GmappingBaseCell (which ignores it) -- the map after each scan appended with that estimator.
Run where oracle/_ref/libslamref.so exists:  python tests/golden/make_golden_omqe.py"""
import os
import sys

import numpy as np

GOLDEN_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(GOLDEN_DIR))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as po  # noqa: E402


def main():
    if not po.ref_available():
        sys.exit("oracle/_ref/libslamref.so missing")
    R = po.Ref()
    scale, n = 0.1, 340
    gt = R.map_create(po.REF_CELL_MOCK, po.MAP_UNBOUNDED_PLAIN, n, n, scale, 0.0)
    gt.stamp_text(R.cecum_text(61, 45, 2), (-30, 20))
    gt.stamp_text(R.cecum_text(25, 17, 3), (-12, -8))
    poses = [(0.05, -0.25, np.deg2rad(90)), (0.45, 0.35, np.deg2rad(60)), (-0.55, -0.15, np.deg2rad(125))]
    steps = [dict(quality=1.0, blur=0.0, max_range=np.inf), dict(quality=0.9, blur=0.3, max_range=np.inf),
             dict(quality=0.7, blur=0.1, max_range=8.0)]
    out = dict(scale=np.array(scale), n_steps=np.array(len(steps)))
    rs = np.random.RandomState(21)
    scans = []
    for k, p in enumerate(poses):
        sc = R.scan_generate(gt, p, 15, 270, 360)
        r, a, o, _ = sc.get()
        o = o.copy()
        miss = rs.rand(r.size) < 0.08
        o[miss] = 0
        r = np.where(miss, 12.0, r + rs.randn(r.size) * 0.005)
        scans.append((r, a, o))
        out["step%d_pose" % k] = np.array(p)
        out["step%d_range" % k], out["step%d_angle" % k], out["step%d_occ" % k] = r, a, o
        out["step%d_params" % k] = np.array([steps[k]["quality"], steps[k]["blur"], steps[k]["max_range"]])
        out["step%d_quality" % k] = R.omqe_quality(R.scan_create(r, a, o))
    models = {"mean": (po.REF_CELL_MEAN, (0.95, 1.0, 0.01, 1.0)), "affine": (po.REF_CELL_AFFINE, (0.95, 1.0, 0.01, 1.0)),
              "tbm": (po.REF_CELL_TBM, (0.95, 0.04, 0.01, 0.003)), "gmapping": (po.REF_CELL_GMAPPING, (0.95, 1.0, 0.01, 1.0))}
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
            R.append_scan(m, R.scan_create(r, a, o), p, quality=steps[k]["quality"], base=base, blur=steps[k]["blur"],
                          max_range=steps[k]["max_range"], omqe=1)
            assert m.geometry() == g0, "the window must not grow in this fixture"
            md = m.to_data()
            out["%s_step%d_payload" % (name, k)] = md.payload[30:310, 30:310].copy()
            aux = m.aux()
            if aux is not None:
                out["%s_step%d_aux" % (name, k)] = aux[30:310, 30:310].copy()
            rest = np.delete(md.payload, np.s_[30:310], axis=0)[..., 0]
            assert np.array_equal(rest, np.full_like(rest, md.unknown[0])), "touched cells outside the crop"
    out["crop"] = np.array([30, 310])
    path = os.path.join(GOLDEN_DIR, "map_update_ahr.npz")
    np.savez_compressed(path, **out)
    print("wrote map_update_ahr.npz", os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
