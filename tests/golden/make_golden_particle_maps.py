#!/usr/bin/env python3
"""G7: per-particle maps pinned to the reference -> tests/golden/particle_maps_cow.npz.

The reference revision never separates the particles' maps (Q20), but its map class has everything the
mode needs: `UnboundedLazyTiledGridMap copy = original` shares every tile until one side writes
(src/core/maps/lazy_tiled_grid_map.h:40-71), which is what `*new_particle = *sampled` does to a
particle's map (src/core/particle_filter.h:92-96).  This script makes REAL copies of a compiled-reference
map and lets the reference scan adder write into them:

  A       ancestor: scan 0 appended from pose 0
  B0..B5  copies of A, scan 1 appended to each from its own pose  (= one batched K6 over six particles)
  C       copy of B2, scan 2 appended                             (= a resampled particle that moves on)
  A, B2   read again afterwards: writes to a copy must not show in the original

with the AreaOccupancyEstimator and blur 0.1 m (slam/occupancy_estimator/type = area, slam/mapping/blur,
BASELINE configs[4]).  Q27 (the estimator's function-local static) is pinned as in make_golden_area.py.

The scans carry the RAW trigonometry provider (what GMapping runs with), for which the reference evaluates
libm sin(theta + a) where the HIP path and its restatement evaluate the angle-addition form: an end point
within an ulp of a cell border can then land in the neighbouring cell (tested on its own).
The reference then amplifies that ulp: are_on_the_same_side (area_occupancy_estimator.h:207-212) tests the
sign of (dy - dx)(x + y) of a cell corner, which is pure rounding noise for corners on the world's
anti-diagonal (x = -y), and flips the obstacle cell between "hit" and "free".  This fixture is about copies and
the batched update, so a pose is redrawn until the restatement evaluated the way the device does (per-beam
libm cos / sin of the scan angle combined with the pose heading by angle addition: a cached-provider scan whose
table holds those values) counts every cell exactly like the reference did with its raw provider."""
import os
import sys

import numpy as np

GOLDEN_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(GOLDEN_DIR))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)
import pyoracle as po  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

PKG = ge.load_package()  # host-side helper only (beam_trig); no GPU is touched
from pyoracle_mapupdate import RULE_GMAPPING, append_scan_ex  # noqa: E402


def main():
    R = po.Ref()
    scale, n = 0.05, 512
    pin = R.map_create(po.REF_CELL_MEAN, po.MAP_UNBOUNDED_PLAIN, 20, 20, scale)
    R.append_scan(pin, R.scan_create([0.005], [0.0]), (scale / 2, scale / 2, 0.0), occ_est=1)
    gt = R.map_create(po.REF_CELL_MOCK, po.MAP_UNBOUNDED_PLAIN, n, n, scale, 0.0)
    gt.stamp_text(R.cecum_text(61, 45, 2), (-30, 20), 2, 2)
    gt.stamp_text(R.cecum_text(25, 17, 3), (-12, -8), 2, 2)
    base, blur = (0.95, 1.0, 0.01, 1.0), 0.1
    pose0 = np.array([scale / 2, scale / 2 - 6 * scale, np.deg2rad(90)])
    rs = np.random.RandomState(3)

    def scan_from(p, noise):
        tp = np.array(p, dtype=np.float64)
        tp[:2] = (np.floor(tp[:2] / scale) + 0.5) * scale
        r, a, o, _ = R.scan_generate(gt, tp, 8, 270, 360).get()
        return r + rs.randn(r.size) * noise, a, o

    scans = [scan_from(pose0, 0.0), scan_from(pose0 + [0.1, 0.05, 0.04], 0.004), scan_from(pose0 + [0.2, 0.1, 0.1], 0.004)]
    A = R.map_create(po.REF_CELL_GMAPPING, po.MAP_UNBOUNDED_LAZY_TILED, n, n, scale, 0.5)
    g0 = A.geometry()
    R.append_scan(A, R.scan_create(*scans[0]), pose0, occ_est=1, base=base, blur=blur)
    out = dict(scale=np.array(scale), origin=np.array(g0["origin"]), size=np.array([g0["width"], g0["height"]]),
               unknown=A.to_data().unknown, base=np.array(base), blur=np.array(blur),
               shift_amount=np.array(0.01 * scale), pose0=pose0)
    for k, (r, a, o) in enumerate(scans):
        out["scan%d_range" % k], out["scan%d_angle" % k], out["scan%d_occ" % k] = r, a, o

    def snap(name, m):
        assert m.geometry() == g0, "the maps must not grow in this fixture"
        out[name + "_payload"] = m.to_data().payload.copy()
        out[name + "_aux"] = m.aux().copy()

    snap("A", A)
    O = po.Oracle()
    shift = 0.01 * scale

    def same_cells(parent, pose, scan):
        """reference copy + append; None unless the restatement's counters equal the reference's"""
        m = parent.copy()
        R.append_scan(m, R.scan_create(*scan), pose, occ_est=1, base=base, blur=blur)
        pd = parent.to_data()
        o_map = po.GridMapData(po.CELL_GMAPPING, pd.payload.copy(), g0["origin"], scale, pd.unknown)
        o_aux = parent.aux().copy()
        # one table slot per beam, holding what slamhip_beam_trig_raw (glibc sincos) hands the device
        cos_a, sin_a = PKG.beam_trig(scan[1])
        dev_trig = po.ScanData(scan[0], scan[1], trig_mode=po.TRIG_CACHED, a_min=0.0, a_delta=1.0,
                               tab_sin=sin_a, tab_cos=cos_a)
        dev_trig.angle = np.arange(scan[1].size, dtype=np.float64)  # the table index of beam i is i
        append_scan_ex(O, o_map, o_aux, RULE_GMAPPING, pose, scan[0], dev_trig.angle, scan[2], base=base, blur=blur,
                       est_kind=1, shift_amount=shift, trig=dev_trig)
        return m if np.array_equal(o_aux, m.aux()) else None

    redrawn = 0
    poses, B = [], []
    for i in range(6):
        while True:
            p = pose0 + [0.1, 0.05, 0.04] + rs.randn(3) * [0.03, 0.03, 0.02]
            b = same_cells(A, p, scans[1])
            if b is not None:
                break
            redrawn += 1
        poses.append(p)
        B.append(b)
    out["poses_b"] = np.array(poses)
    for i, b in enumerate(B):
        snap("B%d" % i, b)
    while True:
        pose_c = pose0 + [0.21, 0.08, 0.11] + rs.randn(3) * [0.01, 0.01, 0.01]
        Cm = same_cells(B[2], pose_c, scans[2])
        if Cm is not None:
            break
        redrawn += 1
    out["pose_c"] = pose_c
    out["poses_redrawn"] = np.array(redrawn)
    snap("C", Cm)
    snap("A_after", A)
    snap("B2_after", B[2])
    assert np.array_equal(out["A_after_payload"], out["A_payload"]) and np.array_equal(out["A_after_aux"], out["A_aux"])
    assert np.array_equal(out["B2_after_payload"], out["B2_payload"]) and not np.array_equal(out["C_aux"], out["B2_aux"])
    del out["A_after_payload"], out["A_after_aux"], out["B2_after_payload"], out["B2_after_aux"]
    path = os.path.join(GOLDEN_DIR, "particle_maps_cow.npz")
    np.savez_compressed(path, **out)
    print("wrote particle_maps_cow.npz", os.path.getsize(path) // 1024, "KiB;", redrawn, "poses redrawn")


if __name__ == "__main__":
    main()
