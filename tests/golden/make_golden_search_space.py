#!/usr/bin/env python3
"""Golden vectors for the brute-force matcher (N1) from the COMPILED REFERENCE:

  bf_smoke.npz      the 8 cases of test/core/scan_matchers/brute_force_sm_smoke_test.cpp:80-118
                    (scene of scan_matcher_test_utils.h, BF 21x21x21): prob, delta, accepted
                    candidates, sha256 of the 9262 scores with the raw and the cached trig provider,
                    full score arrays of two cases
  search_space.npz  the three scenes of src/utils/pose2D_search_space_evaluator.cpp:66-131 (closed
                    corridor, open corridor, several corridors), the 1000-beam scan its
                    run_evaluation generates, and the 201x201 search-space map of
                    ScanMatcherSearchSpaceBuilder (:32-61): BF(-1..1 step 0.01 in x and y, no
                    rotation) scores written to an UnboundedPlainGridMap at 0.01 m, dumped with
                    GridMapToPgmDumber::dump_map.  The reference's own binary cannot be used: it
                    dereferences a null observation-quality estimator in dump_scan
                    (grid_map_scan_adders.h:62 <- pose2D_search_space_evaluator.cpp:150) and
                    segfaults before the first evaluation; the scenes are rebuilt here call by call
                    through oracle/_ref/libslamref.so instead.
  map_growth.npz    UnboundedPlainGridMap::ensure_inside (plain_grid_map.h:133-176): geometry after
                    every update of random and of search-space-ordered cell sequences

    python tests/golden/make_golden_search_space.py
"""
import hashlib
import os
import sys
import tempfile

import numpy as np

GOLDEN_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(GOLDEN_DIR))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from pyoracle import *  # noqa: E402,F401,F403

LEFT, RIGHT, TOP, BOT = 0, 1, 2, 3


def save(name, **kw):
    path = os.path.join(GOLDEN_DIR, name)
    np.savez_compressed(path, **kw)
    print("wrote", name, os.path.getsize(path) // 1024, "KiB")


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


def gen_bf_smoke(R):
    Map_W, Map_H, Scale = 100, 100, 0.1
    m = R.map_create(REF_CELL_MOCK, MAP_UNBOUNDED_PLAIN, Map_W, Map_H, Scale, 0.5)
    cw, ch = 15, 13
    m.stamp_text(R.cecum_text(cw, ch, TOP), (0, 0))
    rpose = np.array([Scale / 2, Scale / 2, 0.0]) + [(cw // 2) * Scale, (-ch + 1) * Scale, np.deg2rad(90)]
    raw = R.scan_generate(m, rpose, 15, 270, 10, 1.0)
    r, a, o, _ = raw.get()
    spe = R.spe_create(OOPE_OBSTACLE, OIE_DISCREPANCY, 0)
    ft, tt, st = -0.5, 0.5, 0.05
    fr, tr, sr = np.deg2rad(-10), np.deg2rad(10), np.deg2rad(1)
    params = [ft, tt, st, ft, tt, st, fr, tr, sr]
    noises = np.array([[0, 0, 0], [ft, 0, 0], [2 * st, 0, 0], [0, -2 * st, 0], [0, tt, 0],
                       [0, 0, -3 * sr], [0, 0, tr], [2 * st, -3 * st, sr]])
    md = m.to_data()
    out = dict(raw_range=r, raw_angle=a, raw_occ=o, rpose=rpose, noises=noises, params=np.array(params),
               map_payload=md.payload, map_origin=np.array(md.origin), map_scale=np.array(md.scale),
               map_unknown=md.unknown, map_cell_model=np.array(md.cell_model), map_bounded=np.array(0))
    hs, inc = np.deg2rad(270 / 2.0), np.deg2rad(270 / 10)
    out["a_min"], out["a_inc"], out["a_max_passed"] = np.array(-hs), np.array(inc), np.array(hs + 2 * inc)
    for i, nz in enumerate(noises):
        for tag, scan in (("raw", R.scan_create(r, a, o)),
                          ("cached", R.scan_create(r, a, o, TRIG_CACHED, -hs, hs + 2 * inc, inc))):
            t = R.process_scan(R.matcher_create(SM_BF, spe, params), scan, rpose + nz, m)
            p = "%s%d_" % (tag, i)
            out[p + "prob"], out[p + "delta"], out[p + "n_calls"] = np.array(t["prob"]), t["delta"], np.array(t["n_calls"])
            out[p + "accepted_idx"] = np.nonzero(t["accepted"])[0].astype(np.int32)
            out[p + "scores_sha256"] = sha(t["scores"])
            out[p + "poses_sha256"] = sha(t["poses"])
            if i in (0, 7):
                out[p + "scores"] = t["scores"]
            if tag == "cached" and i == 0:
                out["tab_sin"], out["tab_cos"] = scan.trig_table()
        fs = R.filter_scan(spe, R.scan_create(r, a, o), rpose, m)
        res_noise = nz + out["raw%d_delta" % i]
        out["case%d_prob_true" % i] = R.score(spe, fs, m, rpose)
        out["case%d_prob_result" % i] = R.score(spe, fs, m, rpose + res_noise)
        print("bf smoke", i, "delta", out["raw%d_delta" % i], "prob", float(out["raw%d_prob" % i]))
    save("bf_smoke.npz", **out)


def scene_closed(R, m):
    m.stamp_text(R.cecum_text(40, 20, RIGHT), (0, 10))
    m.stamp_text(R.cecum_text(20, 40, TOP), (-20, 49))
    m.stamp_text(R.cecum_text(20, 40, BOT), (-20, -9))
    for x in range(-20, 0):
        for y in range(-8, 10):
            m.update(x, y, 0.0, qual=0.0, is_occ=False, quality=1.0)


def scene_open(R, m):
    m.stamp_text(R.cecum_text(80, 20, RIGHT), (-40, 10))
    for y in range(-8, 10):
        m.update(39, y, 0.0, qual=0.0, is_occ=False, quality=1.0)


def scene_several(R, m):
    txt = R.cecum_text(40, 5, RIGHT)
    m.stamp_text(txt, (0, 2))
    m.stamp_text(txt, (0, 7))


def gen_search_space(R):
    R.lib.ref_map_dump_pgm.argtypes = [C.c_void_p, C.c_char_p]
    out = {}
    res = 0.01
    pose = np.array([0.05, 0.05, 0.0])
    params = [-1, 1, res, -1, 1, res, 0, 0, 0.1]
    out["params"], out["pose"], out["resolution"] = np.array(params, dtype=np.float64), pose, np.array(res)
    for name, build in (("closed", scene_closed), ("open", scene_open), ("several", scene_several)):
        m = R.map_create(REF_CELL_MOCK, MAP_UNBOUNDED_PLAIN, 100, 100, 0.1, 0.5)
        build(R, m)
        scan = R.scan_generate(m, pose, 100, 270, 1000, 1.0)
        r, a, o, _ = scan.get()
        spe = R.spe_create(OOPE_OBSTACLE, OIE_DISCREPANCY, 0)
        t = R.process_scan(R.matcher_create(SM_BF, spe, params), scan, pose, m, cap=1 << 17)
        assert t["n_calls"] <= (1 << 17)
        # ScanMatcherSearchSpaceBuilder: last write wins at world_to_cell(pose) of a 0.01 m map
        sss = R.map_create(REF_CELL_MOCK, MAP_UNBOUNDED_PLAIN, 100, 100, res, 0.5)
        cx = np.floor(t["poses"][:, 0] / res).astype(np.int32)
        cy = np.floor(t["poses"][:, 1] / res).astype(np.int32)
        geo = []
        for k in range(t["n_calls"]):
            sss.update(int(cx[k]), int(cy[k]), float(t["scores"][k]), qual=0.0, is_occ=True, quality=1.0)
            if k % 997 == 0 or k == t["n_calls"] - 1:
                g = sss.geometry()
                geo.append([k, g["width"], g["height"], g["origin"][0], g["origin"][1]])
        with tempfile.TemporaryDirectory() as td:
            for tag, mm in (("input", m), ("sss", sss)):
                R.lib.ref_map_dump_pgm(mm.h, os.path.join(td, tag + ".pgm").encode())
                out["%s_%s_pgm" % (name, tag)] = np.frombuffer(open(os.path.join(td, tag + ".pgm"), "rb").read(), np.uint8)
        md, sd = m.to_data(), sss.to_data()
        out[name + "_map_payload"], out[name + "_map_origin"] = md.payload, np.array(md.origin)
        out[name + "_scan"] = np.stack([r, a, o.astype(np.float64)])
        out[name + "_n_calls"], out[name + "_prob"], out[name + "_delta"] = np.array(t["n_calls"]), np.array(t["prob"]), t["delta"]
        out[name + "_accepted_idx"] = np.nonzero(t["accepted"])[0].astype(np.int32)
        out[name + "_scores_sha256"] = sha(t["scores"])
        out[name + "_sss_geometry"] = np.array(geo, dtype=np.int32)
        out[name + "_sss_final"] = np.array([sd.width, sd.height, sd.origin[0], sd.origin[1]], dtype=np.int32)
        if name == "closed":
            out[name + "_scores"] = t["scores"]
        else:  # every 4th candidate keeps the fixture small
            out[name + "_scores_every4"] = t["scores"][::4].copy()
        print(name, "calls", t["n_calls"], "best", t["prob"], "delta", t["delta"], "sss", sd.width, sd.height, sd.origin)
    save("search_space.npz", **out)


def gen_map_growth(R):
    rs = np.random.RandomState(5)
    out = {}
    for k, (w, h) in enumerate([(100, 100), (10, 7), (1, 1), (64, 200)]):
        m = R.map_create(REF_CELL_MOCK, MAP_UNBOUNDED_PLAIN, w, h, 0.1, 0.5)
        g0 = m.geometry()
        # mixture of small steps outside, big jumps, and cells already inside
        cells = np.cumsum(rs.randint(-9, 10, size=(60, 2)) * rs.choice([1, 1, 1, 7], size=(60, 1)), axis=0)
        cells[::5] = rs.randint(-400, 400, size=cells[::5].shape)
        geo = []
        for c in cells:
            m.update(int(c[0]), int(c[1]), 0.7)
            g = m.geometry()
            geo.append([g["width"], g["height"], g["origin"][0], g["origin"][1]])
        out["seq%d_start" % k] = np.array([g0["width"], g0["height"], g0["origin"][0], g0["origin"][1]], dtype=np.int32)
        out["seq%d_cells" % k] = cells.astype(np.int32)
        out["seq%d_geometry" % k] = np.array(geo, dtype=np.int32)
    save("map_growth.npz", **out)


if __name__ == "__main__":
    R = Ref()
    gen_map_growth(R)
    gen_bf_smoke(R)
    gen_search_space(R)
