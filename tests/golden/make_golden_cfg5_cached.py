#!/usr/bin/env python3
"""cfg5's geometry pinned to the compiled reference with the CACHED trigonometry provider
-> tests/golden/cfg5_cached.npz  (VERDICT r2 item 1).

BASELINE configs[4]: 0.025 m cells, 1080 beams over 270 degrees whose walks are hundreds of cells long,
AreaOccupancyEstimator, blur 0.1 m (four cells), GmappingBaseCell maps in an UnboundedLazyTiledGridMap.
tests/golden/particle_maps_cow.npz and the cfg5 oracle test run the RAW provider, where the reference takes libm
sin(theta + a) and the device the angle-addition form: an occasional beam grazing a cell corner then counts
differently, and those tests have to explain such cells away.  With CachedTrigonometryProvider
(src/core/trigonometry_utils.h:45-78) the reference itself evaluates cos_b cos_a - sin_b sin_a from a table of the
beam angles -- exactly the device's arithmetic -- so EVERY cell must agree: hit / try counters exact, payload to
1e-10, no exceptions.

Two particles' histories, three scans each, from poses a few centimetres apart:
  P  scan 0, 1, 2 from poses_p[0..2]  -> snapshots P0 (fresh map, one scan) and P2 (three scans)
  Q  scan 0, 1, 2 from poses_q[0..2]  -> snapshot Q2
Ranges come from tests/synth.py's ray caster on its synthetic world (scans are inputs; the reference's own
LaserScanGenerator only returns the beams that hit something, with its own angle convention).
The scans' angles ARE the table's angles (the accumulating `angle += delta` loop of the provider's update), so the
table entry of beam i is libm's value at angle[i] -- which is what slamhip_beam_trig_raw gives the device.
Snapshots are stored sparsely (cells that differ from the never-observed prototype): flat index deltas + values.
Q27 (the estimator's function-local static shift) is pinned as in make_golden_area.py."""
import os
import sys

import numpy as np

GOLDEN_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(GOLDEN_DIR))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)
import pyoracle as po  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "tests"))
from synth import cast_scan, make_world  # noqa: E402

PKG = ge.load_package()  # host-side helper only (beam_trig); no GPU is touched


def sparse(payload, aux, unknown):
    """cells that were ever written: (flat index deltas, payload rows, counter rows)"""
    touched = (aux != 0).any(-1) | (payload != np.asarray(unknown)[None, None, :payload.shape[-1]]).any(-1)
    flat = np.flatnonzero(touched)
    return (np.diff(flat, prepend=0).astype(np.uint32), payload.reshape(-1, payload.shape[-1])[flat],
            aux.reshape(-1, 2)[flat].astype(np.uint16))


def main():
    R = po.Ref()
    scale, n = 0.025, 1280  # a 32 m world
    pin = R.map_create(po.REF_CELL_MEAN, po.MAP_UNBOUNDED_PLAIN, 20, 20, scale)
    R.append_scan(pin, R.scan_create([0.005], [0.0]), (scale / 2, scale / 2, 0.0), occ_est=1)
    gt = make_world(n, scale, 6)  # tests/synth.py: rooms and pillars on a raster, origin at its centre
    base, blur, beams, fov = (0.95, 1.0, 0.01, 1.0), 0.1, 1080, 270
    # the provider's table: angle += delta from a_min while angle < a_max (laser_scan_observer.h:80 passes a_max + inc)
    inc = np.deg2rad(fov) / beams
    a_min = -np.deg2rad(fov) / 2
    a_max = a_min + inc * beams + inc
    O = po.Oracle()
    tab_sin, tab_cos = O.trig_table(a_min, a_max, inc)
    ang, a = [], a_min
    while a < a_max:
        ang.append(a)
        a += inc
    ang = np.array(ang[:beams])
    tab_sin, tab_cos = tab_sin[:ang.size + 1], tab_cos[:ang.size + 1]
    dev_cos, dev_sin = PKG.beam_trig(ang)
    assert np.array_equal(dev_cos, tab_cos[:beams]) and np.array_equal(dev_sin, tab_sin[:beams]), \
        "the device's per-beam trig is not the provider's table"
    rs = np.random.RandomState(11)
    pose0 = np.array([scale / 2, scale / 2, np.deg2rad(72)])
    poses_p = pose0 + np.array([[0, 0, 0], [0.11, 0.06, 0.03], [0.23, 0.09, 0.08]])
    poses_q = poses_p + rs.randn(3, 3) * [0.04, 0.04, 0.015]

    def ranges_from(p, seed):
        # tests/synth.py's ray caster on that world (beams that hit nothing are dropped), N(0, 0.01 m) range noise
        # (SURVEY 8d); the kept beams take the TABLE's angles, whose spacing the caster's differs from by an ulp
        r, a_cast = cast_scan(gt, scale, p, beams, fov_deg=fov, max_dist=12.0, noise=0.01, seed=seed)
        idx = np.rint((a_cast - a_min) / inc).astype(np.int64)
        assert np.abs(a_cast - ang[idx]).max() < 1e-9 and np.unique(idx).size == idx.size
        return np.clip(r, 0.05, None), idx

    scans = [ranges_from(p, 40 + k) for k, p in enumerate(poses_p)]

    def history(poses, keep):
        m = R.map_create(po.REF_CELL_GMAPPING, po.MAP_UNBOUNDED_LAZY_TILED, n, n, scale, 0.5)
        g0 = m.geometry()
        snaps = {}
        for k, (p, (r, idx)) in enumerate(zip(poses, scans)):
            sc = R.scan_create(r, ang[idx], None, po.TRIG_CACHED, a_min, a_max, inc)
            R.append_scan(m, sc, p, occ_est=1, base=base, blur=blur)
            assert m.geometry() == g0, "the map must not grow in this fixture"
            if k in keep:
                snaps[k] = (m.to_data().payload.copy(), m.aux().copy())
        return g0, m.to_data().unknown, snaps

    g0, unknown, sp = history(poses_p, (0, 2))
    _, _, sq = history(poses_q, (2,))
    out = dict(scale=np.array(scale), origin=np.array(g0["origin"]), size=np.array([g0["width"], g0["height"]]),
               unknown=unknown, base=np.array(base), blur=np.array(blur), shift_amount=np.array(0.01 * scale),
               angle=ang, a_min=np.array(a_min), a_max=np.array(a_max), a_inc=np.array(inc), poses_p=poses_p,
               poses_q=poses_q)
    for k, (r, idx) in enumerate(scans):
        out["scan%d_range" % k], out["scan%d_beam" % k] = r, idx.astype(np.int32)  # beam i looks along angle[beam[i]]
    cells = 0
    for name, (pay, aux) in (("P0", sp[0]), ("P2", sp[2]), ("Q2", sq[2])):
        assert aux.max() < 65536
        d, v, c = sparse(pay, aux, unknown)
        out[name + "_idx_delta"], out[name + "_payload"], out[name + "_counters"] = d, v, c
        cells += d.size
    assert int((sp[2][1][..., 1] - sp[0][1][..., 1]).max()) >= 2  # cells met again by the later scans
    a0 = ang[scans[0][1]]
    walk = np.abs(np.floor((poses_p[0][0] + scans[0][0] * np.cos(poses_p[0][2] + a0)) / scale) - np.floor(poses_p[0][0] / scale)) + \
        np.abs(np.floor((poses_p[0][1] + scans[0][0] * np.sin(poses_p[0][2] + a0)) / scale) - np.floor(poses_p[0][1] / scale))
    path = os.path.join(GOLDEN_DIR, "cfg5_cached.npz")
    np.savez_compressed(path, **out)
    print("wrote cfg5_cached.npz", os.path.getsize(path) // 1024, "KiB;", cells, "cells in three snapshots; walks of up to",
          int(walk.max()), "cells, mean", int(walk.mean()))


if __name__ == "__main__":
    main()
