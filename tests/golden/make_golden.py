#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the COMPILED REFERENCE.

Runs only where oracle/_ref/libslamref.so exists (the build container: `make -C oracle ref`
compiles the unmodified reference headers under /root/reference in place).  The fixtures are
data only -- inputs and the reference's outputs -- and are committed so that the CPU and GPU test
suites can check the oracle and the HIP path without the reference being present.

    python tests/golden/make_golden.py

Fixtures written (np.savez_compressed):
  enumerators.npz   pose-enumerator known answers (SURVEY Appendix B pins + HC(6), seeds)
  oope_known.npz    the 18 known-answer cases of
                    test/core/scan_matchers/occupancy_observation_probability_test.cpp:59-207
                    (inputs, the expected literals of that file, and the reference's outputs)
  scene_<cell>_<trig>.npz  G1/G2/G3/G4: map window payload, raw + filtered scan, weights,
                    per-pose scores, MC and HC accept traces (cell in mean|tbm|affine,
                    trig in raw|cached)
  hc_smoke.npz      the 7 cases of test/core/scan_matchers/hill_climbing_sm_smoke_test.cpp:72-105
  gmapping_scene.npz G1 for the GMapping 3x3 OOPE incl. the run-cache quirk, + HC(6,0.1,0.1) trace
  gmapping_pf.npz   G5: multi-step GmappingParticleFilter runs (poses, weights, master flags,
                    resampling decisions per step) with seeds injected
  resample.npz      G5: weights + seeds -> N_eff decision and resampling indices
  weights_ahr.npz   G4: angle-histogram weights on a noisy scan
  world_to_cells.npz  A16 ray-walk cell lists for random + axis-aligned + diagonal segments
"""
import os
import sys

import numpy as np

GOLDEN_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(GOLDEN_DIR))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from pyoracle import *  # noqa: E402,F401,F403


def save(name, **kw):
    path = os.path.join(GOLDEN_DIR, name)
    np.savez_compressed(path, **kw)
    print("wrote", name, os.path.getsize(path) // 1024, "KiB")


def map_fields(md, prefix="map_"):
    return {prefix + "payload": md.payload, prefix + "origin": np.array(md.origin),
            prefix + "scale": np.array(md.scale), prefix + "unknown": md.unknown,
            prefix + "cell_model": np.array(md.cell_model), prefix + "bounded": np.array(int(md.bounded))}


def crop(md, margin_cells, ext_lo, ext_hi):
    """Crop a GridMapData to external cells [ext_lo-margin, ext_hi+margin] (keeps fixtures small)."""
    ox, oy = md.origin
    x0 = max(0, ext_lo[0] - margin_cells + ox)
    y0 = max(0, ext_lo[1] - margin_cells + oy)
    x1 = min(md.width, ext_hi[0] + margin_cells + ox + 1)
    y1 = min(md.height, ext_hi[1] + margin_cells + oy + 1)
    return GridMapData(md.cell_model, md.payload[y0:y1, x0:x1].copy(), (ox - x0, oy - y0), md.scale,
                       md.unknown, md.bounded)


def trace_fields(t, prefix):
    return {prefix + "prob": np.array(t["prob"]), prefix + "delta": t["delta"],
            prefix + "n_calls": np.array(t["n_calls"]), prefix + "poses": t["poses"],
            prefix + "scores": t["scores"], prefix + "accepted": t["accepted"]}


def build_world(R, scale=0.1, n=200):
    """Ground truth: two nested cecum primitives (map_primitives.h:64-154) on a MockGridCell map."""
    gt = R.map_create(REF_CELL_MOCK, MAP_UNBOUNDED_PLAIN, n, n, scale, 0.0)
    gt.stamp_text(R.cecum_text(61, 45, 2), (-30, 20))
    gt.stamp_text(R.cecum_text(25, 17, 3), (-12, -8))
    pose = (scale / 2, scale / 2 - 3 * scale, np.deg2rad(90))
    return gt, pose


def gen_enumerators(R):
    out = {}
    cases = {"mc_666666": (SM_MC, [666666, 0.2, 0.1, 20, 100], [0, 0, 0]),
             "mc_42": (SM_MC, [42, 0.05, 0.3, 7, 50], [1.5, -2.25, 0.7]),
             "hc_2": (SM_HC, [2, 0.1, 0.2], [0, 0, 0]),
             "hc_6": (SM_HC, [6, 0.1, 0.1], [0.3, -0.2, 0.1])}
    for k, (kind, p, base) in cases.items():
        out[k + "_kind"] = np.array(kind)
        out[k + "_params"] = np.array(p, dtype=np.float64)
        out[k + "_base"] = np.array(base, dtype=np.float64)
        out[k + "_poses"] = R.enumerate_all_rejected(kind, p, base)
    save("enumerators.npz", **out)


def gen_oope_known(R):
    """occupancy_observation_probability_test.cpp: 2x2 patch on a 100x100 @1.0 map of
    MockGridCell(0.0); expected literals are the numbers written in that test file."""
    m = R.map_create(REF_CELL_MOCK, MAP_UNBOUNDED_PLAIN, 100, 100, 1.0, 0.0)
    for (x, y), v in {(0, 1): 0.25, (1, 1): 0.0, (0, 0): 1.0, (1, 0): 0.5}.items():
        m.update(x, y, v, 1.0)
    S = 1.0
    mid = lambda i, j: (S / 2 + i * S, S / 2 + j * S)  # noqa: E731
    cell = (0.0, S, 0.0, S)  # bot, top, left, right

    def shrink(r, f):
        cy, cx = r[0] + (r[1] - r[0]) / 2, r[2] + (r[3] - r[2]) / 2
        hv, hh = (r[1] - r[0]) / (f * 2), (r[3] - r[2]) / (f * 2)
        return (cy - hv, cy + hv, cx - hh, cx + hh)

    def sh(p, dx, dy, k=1.0):
        return (p[0] + dx * k, p[1] + dy * k)

    pt = (0.0, 0.0, 0.0, 0.0)
    h = shrink(cell, 2)
    cases = [
        (OOPE_OBSTACLE, 1, mid(0, 0), pt), (OOPE_OBSTACLE, 1, mid(0, 0), (-1000, 1000, -1000, 1000)),
        (OOPE_OBSTACLE, 1, sh(mid(0, 0), S, S, 0.25), pt), (OOPE_OBSTACLE, 0.25, sh(mid(0, 0), 0, S), pt),
        (OOPE_OBSTACLE, 0.50, sh(mid(0, 0), S, 0), pt), (OOPE_OBSTACLE, 0.0, sh(mid(0, 0), S, S), pt),
        (OOPE_MAX, 1, mid(0, 0), pt), (OOPE_MAX, 0.25, mid(0, 1), h),
        (OOPE_MAX, 1, sh(mid(0, 1), 0, -S, 0.5), h), (OOPE_MAX, 0.25, sh(mid(0, 1), S, 0, 0.5), h),
        (OOPE_MAX, 1, sh(mid(0, 1), S, -S, 0.5), h), (OOPE_MAX, 0.5, sh(mid(1, 0), 0, S, 0.5), h),
        (OOPE_MAX, 1, sh(mid(0, 1), 0, -S, 0.375), h), (OOPE_MAX, 0.25, sh(mid(0, 1), S, 0, 0.375), h),
        (OOPE_MAX, 1, sh(mid(0, 1), S, -S, 0.375), h),
        (OOPE_MAX, 0.0, mid(1, -1), shrink(cell, 2)), (OOPE_MAX, 0.5, mid(1, -1), cell),
        (OOPE_MAX, 1, mid(1, -1), shrink(cell, 0.5)),
        (OOPE_MEAN, 1, mid(0, 0), pt), (OOPE_MEAN, 0.25, mid(0, 1), h),
        (OOPE_MEAN, 1.25 / 2, sh(mid(0, 1), 0, -S, 0.5), h), (OOPE_MEAN, 0.25 / 2, sh(mid(0, 1), S, 0, 0.5), h),
        (OOPE_MEAN, 1.75 / 4, sh(mid(0, 1), S, -S, 0.5), h),
        (OOPE_MEAN, 1.25 / 2, sh(mid(0, 1), 0, -S, 0.375), h), (OOPE_MEAN, 0.25 / 2, sh(mid(0, 1), S, 0, 0.375), h),
        (OOPE_MEAN, 1.75 / 4, sh(mid(0, 1), S, -S, 0.375), h), (OOPE_MEAN, 0.5 / 2, sh(mid(1, 0), 0, S, 0.375), h),
        (OOPE_MEAN, 0.0, mid(1, -1), shrink(cell, 2)), (OOPE_MEAN, 0.5 / 4, mid(1, -1), cell),
        (OOPE_MEAN, 1.5 / 9, mid(1, -1), shrink(cell, 0.5)),
        (OOPE_OVERLAP, 1, mid(0, 0), pt), (OOPE_OVERLAP, 0.25, mid(0, 1), h),
        (OOPE_OVERLAP, 0.625, sh(mid(0, 1), 0, -S, 0.5), h), (OOPE_OVERLAP, 0.25 / 2, sh(mid(0, 1), S, 0, 0.5), h),
        (OOPE_OVERLAP, 1.75 / 4, sh(mid(0, 1), S, -S, 0.5), h), (OOPE_OVERLAP, 0.5 / 2, sh(mid(1, 0), 0, S, 0.5), h),
        (OOPE_OVERLAP, 0.4375, sh(mid(0, 1), 0, -S, 0.375), h), (OOPE_OVERLAP, 0.1875, sh(mid(0, 1), S, 0, 0.375), h),
        (OOPE_OVERLAP, 0.25 * 0.25 * 0.5 + 0.25 * 0.75 * 1 + 0.75 * 0.75 * 0.25, sh(mid(0, 1), S, -S, 0.375), h),
        (OOPE_OVERLAP, 0.5 * 0.25 + 1 * 0.125 + 0.25 * 0.0625, mid(1, 0), shrink(cell, 0.5)),
        (OOPE_OVERLAP, (0.5 * 1 + 1 * 0.25 + 0.25 * 0.0625) / 2.25, mid(1, 0), shrink(cell, 2.0 / 3.0)),
        (OOPE_OBSTACLE, 0.5, mid(1, 0), pt), (OOPE_MAX, 0.5, mid(1, 0), pt),
        (OOPE_MEAN, 0.5, mid(1, 0), pt), (OOPE_OVERLAP, 0.5, mid(1, 0), pt),
    ]
    kinds, exp, obst, rng, got = [], [], [], [], []
    for k, e, o, r in cases:
        kinds.append(k)
        exp.append(e)
        obst.append(o)
        rng.append(r)
        got.append(R.oope_probability(k, OIE_DISCREPANCY, m, o[0], o[1], r))
    exp, got = np.array(exp, dtype=np.float64), np.array(got)
    assert np.all(np.abs(exp - got) <= np.finfo(np.float64).eps), (exp - got)
    md = crop(m.to_data(), 4, (-2, -2), (3, 3))
    save("oope_known.npz", kinds=np.array(kinds), expected_literal=exp, obstacle=np.array(obst),
         range4=np.array(rng), reference_out=got, **map_fields(md))


def gen_scene(R, cell, cell_name, weighting, wname, trig, trig_name, base, n_beams=720):
    gt, pose = build_world(R)
    raw = R.scan_generate(gt, pose, 15, 270, n_beams)
    r, a, o, _ = raw.get()
    m = R.map_create(cell, MAP_UNBOUNDED_PLAIN, 200, 200, 0.1)
    for _k in range(5):
        R.append_scan(m, raw, pose, quality=0.9, base=base, blur=0.3)
    md = crop(m.to_data(), 40, (-35, -30), (35, 30))
    inc = a[1] - a[0]
    a_min, a_max = a[0], a[-1] + inc
    scan = R.scan_create(r, a, o, trig, a_min, a_max + inc, inc)
    spe = R.spe_create(OOPE_OBSTACLE, OIE_DISCREPANCY, weighting)
    noisy = np.array([pose[0] + 0.07, pose[1] - 0.04, pose[2] + 0.03])
    fs = R.filter_scan(spe, scan, noisy, m)
    fr, fa, _fo, ff = fs.get()
    wts = R.scan_weights(spe, fs)
    ts, tc = scan.trig_table()
    rs = np.random.RandomState(20260101)
    poses = noisy + rs.randn(96, 3) * [0.2, 0.2, 0.1]
    poses[0] = noisy
    poses[1] = noisy  # identical poses must give bit-identical scores (tie consistency)
    poses[2] = [40.0, 40.0, 0.3]  # every endpoint outside the window -> unknown cell
    scores = R.score(spe, fs, m, poses)
    out = dict(raw_range=r, raw_angle=a, raw_occ=o, a_min=np.array(a_min), a_inc=np.array(inc),
               a_max_passed=np.array(a_max + inc), trig_mode=np.array(trig), tab_sin=ts, tab_cos=tc,
               weighting=np.array(weighting), init_pose=noisy, true_pose=np.array(pose),
               f_range=fr, f_angle=fa, f_factor=ff, f_weight=wts, poses=poses, scores=scores,
               **map_fields(md))
    for name, kind, p in [("mc", SM_MC, [666666, 0.2, 0.1, 20, 100]),
                          ("mc_long", SM_MC, [4242, 0.2, 0.1, 60, 300]),
                          ("hc6", SM_HC, [6, 0.1, 0.1]), ("hc128", SM_HC, [128, 0.1, 0.1])]:
        mt = R.matcher_create(kind, spe, p)
        t = R.process_scan(mt, scan, noisy, m)
        out[name + "_kind"] = np.array(kind)
        out[name + "_params"] = np.array(p, dtype=np.float64)
        out.update(trace_fields(t, name + "_"))
        if name == "mc":  # second call on the same matcher: engine is NOT reseeded (Q7)
            t2 = R.process_scan(mt, scan, noisy, m)
            out.update(trace_fields(t2, "mc_second_"))
    # max / mean / overlap window OOPEs on the same scene (K2)
    area = (0.0, 0.25, 0.0, 0.25)
    for kname, kk in [("max", OOPE_MAX), ("mean", OOPE_MEAN), ("overlap", OOPE_OVERLAP)]:
        spe2 = R.spe_create(kk, OIE_DISCREPANCY, weighting)
        fs2 = R.filter_scan(spe2, scan, noisy, m)
        out["win_" + kname + "_scores"] = R.score(spe2, fs2, m, poses[:24], area)
    out["win_area"] = np.array(area)
    save("scene_%s_%s.npz" % (cell_name, trig_name), **out)


def gen_hc_smoke(R):
    """hill_climbing_sm_smoke_test.cpp:72-105 + scan_matcher_test_utils.h:46-80."""
    Map_W, Map_H, Scale = 100, 100, 0.1
    m = R.map_create(REF_CELL_MOCK, MAP_UNBOUNDED_PLAIN, Map_W, Map_H, Scale, 0.5)
    cw, ch = 15, 13
    m.stamp_text(R.cecum_text(cw, ch, 2), (0, 0))
    rpose = np.array([Scale / 2, Scale / 2, 0.0]) + [(cw // 2) * Scale, (-ch + 1) * Scale, np.deg2rad(90)]
    raw = R.scan_generate(m, rpose, 15, 270, 10, 1.0)
    r, a, o, _ = raw.get()
    spe = R.spe_create(OOPE_OBSTACLE, OIE_DISCREPANCY, 0)
    step, ang = 0.1, np.deg2rad(30)
    noises = np.array([[0, 0, 0], [-step, 0, 0], [step, 0, 0], [0, -step, 0], [0, step, 0],
                       [0, 0, ang], [0, 0, -ang]])
    md = crop(m.to_data(), 30, (0, -13), (15, 0))
    out = dict(raw_range=r, raw_angle=a, raw_occ=o, rpose=rpose, noises=noises,
               params=np.array([10, step, ang]), **map_fields(md))
    # the generator's angles accumulate from -hsector by inc (laser_scan_generator.h:47), exactly
    # like CachedTrigonometryProvider::update builds its table, so the cached provider is usable
    hs, inc = np.deg2rad(270 / 2.0), np.deg2rad(270 / 10)
    out["a_min"], out["a_inc"], out["a_max_passed"] = np.array(-hs), np.array(inc), np.array(hs + 2 * inc)
    for i, nz in enumerate(noises):
        mt = R.matcher_create(SM_HC, spe, [10, step, ang])
        scan = R.scan_create(r, a, o)
        t = R.process_scan(mt, scan, rpose + nz, m)
        out.update(trace_fields(t, "case%d_" % i))
        fs = R.filter_scan(spe, scan, rpose, m)
        res_noise = nz + t["delta"]
        out["case%d_prob_true" % i] = R.score(spe, fs, m, rpose)
        out["case%d_prob_result" % i] = R.score(spe, fs, m, rpose + res_noise)
        # same case with the reference's CachedTrigonometryProvider (use_trig_cache=true)
        mt = R.matcher_create(SM_HC, spe, [10, step, ang])
        cscan = R.scan_create(r, a, o, TRIG_CACHED, -hs, hs + 2 * inc, inc)
        out.update(trace_fields(R.process_scan(mt, cscan, rpose + nz, m), "cached%d_" % i))
        if i == 0:
            out["tab_sin"], out["tab_cos"] = cscan.trig_table()
    save("hc_smoke.npz", **out)


def gen_gmapping_scene(R):
    gt, pose = build_world(R, scale=0.05, n=400)
    raw = R.scan_generate(gt, pose, 15, 270, 720)
    r, a, o, _ = raw.get()
    m = R.map_create(REF_CELL_GMAPPING, MAP_UNBOUNDED_LAZY_TILED, 400, 400, 0.05)
    rs = np.random.RandomState(7)
    for _k in range(5):
        jit = np.array(pose) + rs.randn(3) * [0.01, 0.01, 0.002]
        R.append_scan(m, raw, jit, quality=1.0, base=(0.95, 1.0, 0.01, 1.0), blur=0.0)
    full = m.to_data()
    md = crop(full, 60, (-35, -30), (35, 30))
    scan = R.scan_create(r, a, o)
    spe = R.spe_create(OOPE_GMAPPING, OIE_DISCREPANCY, 0, skip_rate=0)
    noisy = np.array([pose[0] + 0.03, pose[1] - 0.02, pose[2] + 0.01])
    fs = R.filter_scan(spe, scan, noisy, m)
    fr, fa, _fo, ff = fs.get()
    wts = R.scan_weights(spe, fs)
    poses = noisy + rs.randn(64, 3) * [0.05, 0.05, 0.02]
    poses[0] = noisy
    poses[1] = noisy
    # ONE spe (one OOPE cache) scores all poses in order: cache carries across poses (Q19)
    scores = R.score(spe, fs, m, poses)
    out = dict(raw_range=r, raw_angle=a, raw_occ=o, init_pose=noisy, true_pose=np.array(pose),
               f_range=fr, f_angle=fa, f_factor=ff, f_weight=wts, poses=poses, scores=scores,
               **map_fields(md))
    spe2 = R.spe_create(OOPE_GMAPPING, OIE_DISCREPANCY, 0, skip_rate=3)
    mt = R.matcher_create(SM_HC, spe2, [6, 0.1, 0.1])
    t = R.process_scan(mt, scan, noisy, m)
    out.update(trace_fields(t, "hc6_skip3_"))
    fs3 = R.filter_scan(spe2, scan, noisy, m)
    out["skip3_range"], out["skip3_angle"] = fs3.get()[:2]
    save("gmapping_scene.npz", **out)


def gen_gmapping_pf(R):
    """G5: multi-step GmappingParticleFilter runs of the compiled reference with the map update
    switched off through the reference's own parameter (slam/mapping/max_range = 0 makes
    WallDistanceBlurringScanAdder::handle_scan_point return at once, grid_map_scan_adders.h:140-142),
    so that particles only read the (pre-built) shared map."""
    from pyoracle import RefGmapping
    scale, n_cells = 0.05, 400
    gt, pose0 = build_world(R, scale=scale, n=n_cells)
    out = {}
    scenarios = {
        # default GMapping parameters (init_gmapping.h:15-34): big rotations pass the matching gate
        "default": dict(gp=[0.0, 0.1, 0.0, 0.03, 0.6, 0.8, 0.3, 0.4],
                        deltas=[[0.0, 0.0, 0.0], [0.02, 0.01, 0.45], [0.03, -0.02, -0.47],
                                [0.01, 0.02, 0.5], [-0.02, 0.0, -0.44], [0.0, 0.01, 0.46]], n=12),
        # gate forced open (sm_delta_lim min = max = 0, SURVEY 8d metric 2), small motions
        "nogate": dict(gp=[0.0, 0.1, 0.0, 0.03, 0.0, 0.0, 0.0, 0.0],
                       deltas=[[0.0, 0.0, 0.0], [0.15, 0.05, 0.05], [0.2, -0.1, 0.08], [0.25, 0.1, -0.1],
                               [0.1, 0.2, 0.12], [0.3, 0.05, 0.1], [0.2, 0.1, 0.05]], n=20),
        # wide pose noise: weights diverge, N_eff drops and resampling (with duplicated particles,
        # master hand-over) happens
        "wide": dict(gp=[0.0, 0.3, 0.0, 0.12, 0.0, 0.0, 0.0, 0.0],
                     deltas=[[0.0, 0.0, 0.0], [0.3, 0.2, 0.1], [0.35, -0.2, 0.15], [0.3, 0.3, -0.2],
                             [0.4, 0.2, 0.25], [0.3, -0.3, 0.2], [0.35, 0.25, -0.15], [0.3, 0.3, 0.2],
                             [0.4, -0.2, 0.25]], n=16),
    }
    for name, sc in scenarios.items():
        n = sc["n"]
        seeds = np.arange(1000, 1000 + n, dtype=np.uint32)
        g = RefGmapping(R, n, n_cells, n_cells, scale, sc["gp"], seeds, skip_rate=3, map_max_range=0.0)
        mview = g.map()
        raw = R.scan_generate(gt, pose0, 15, 270, 720)
        rs = np.random.RandomState(77)
        for _k in range(5):
            jit = np.array(pose0) + rs.randn(3) * [0.01, 0.01, 0.002]
            R.append_scan(mview, raw, jit, quality=1.0, blur=0.0)
        md = crop(mview.to_data(), 80, (-35, -30), (35, 30))
        out.update({name + "_" + k: v for k, v in map_fields(md).items()})
        out[name + "_gp"] = np.array(sc["gp"])
        out[name + "_seeds"] = seeds
        out[name + "_deltas"] = np.array(sc["deltas"])
        true = np.zeros(3)  # particles start at the origin; the robot's true pose is pose0 + odometry
        for k, d in enumerate(sc["deltas"]):
            true = true + np.array(d)
            tp = np.array(pose0) + true - np.array([0, 0, 0])
            # cell-boundary poses are rejected by the generator: nudge by a fraction of a cell
            tp[:2] = (np.floor(tp[:2] / scale) + 0.5) * scale
            scan = R.scan_generate(gt, tp, 15, 270, 720)
            r, a, o, _ = scan.get()
            # the filter believes odometry: first delta places the particles at pose0
            dd = np.array(pose0) if k == 0 else np.array(d)
            extra = np.arange(5000 + 100 * k, 5000 + 100 * k + n, dtype=np.uint32)
            res, poses, w, ms = g.step(scan, dd, 7 + k, extra)
            out["%s_step%d_range" % (name, k)] = r
            out["%s_step%d_angle" % (name, k)] = a
            out["%s_step%d_delta" % (name, k)] = dd
            out["%s_step%d_resampled" % (name, k)] = np.array(int(res))
            out["%s_step%d_poses" % (name, k)] = poses
            out["%s_step%d_weights" % (name, k)] = w
            out["%s_step%d_master" % (name, k)] = ms
        out[name + "_n_steps"] = np.array(len(sc["deltas"]))
    save("gmapping_pf.npz", **out)


def gen_resample(R):
    rs = np.random.RandomState(5)
    out = {}
    for k, n in enumerate([1, 2, 10, 100, 500]):
        w = rs.rand(n) ** (1 + 3 * (k % 2))
        w /= w.sum()
        req, idx = R.resample(w, 7 + k)
        out["w%d" % k], out["seed%d" % k] = w, np.array(7 + k)
        out["req%d" % k], out["idx%d" % k] = np.array(int(req)), idx
    w = np.zeros(16)
    w[3] = 1.0  # degenerate
    req, idx = R.resample(w, 99)
    out["w5"], out["seed5"], out["req5"], out["idx5"] = w, np.array(99), np.array(int(req)), idx
    w = rs.rand(50) * 0.01  # un-normalised, sum < 1: indices stay 0 when u >= sum (Q24)
    req, idx = R.resample(w, 123)
    out["w6"], out["seed6"], out["req6"], out["idx6"] = w, np.array(123), np.array(int(req)), idx
    out["n_cases"] = np.array(7)
    save("resample.npz", **out)


def gen_weights_ahr(R):
    rs = np.random.RandomState(11)
    n = 360
    a = np.deg2rad(np.linspace(-135, 135, n))
    r = 4 + np.sin(a * 3) + rs.rand(n) * 0.3
    m = R.map_create(REF_CELL_MEAN, MAP_UNBOUNDED_PLAIN, 400, 400, 0.1)
    scan = R.scan_create(r, a)
    out = dict(range=r, angle=a)
    for name, k in [("even", 0), ("viny", 1), ("ahr", 2)]:
        spe = R.spe_create(OOPE_OBSTACLE, OIE_DISCREPANCY, k)
        fs = R.filter_scan(spe, scan, (0.0, 0.0, 0.0), m)
        assert fs.size() == n
        out["w_" + name] = R.scan_weights(spe, fs)
    # skip_rate / max_range filter (Q8, Q10)
    occ = (rs.rand(n) > 0.1).astype(np.int32)
    scan2 = R.scan_create(r, a, occ)
    spe = R.spe_create(OOPE_OBSTACLE, OIE_DISCREPANCY, 0, skip_rate=3, max_range=4.5)
    fs = R.filter_scan(spe, scan2, (0.3, -0.2, 0.4), m)
    out["occ"], out["filt_pose"] = occ, np.array([0.3, -0.2, 0.4])
    out["filt_range"], out["filt_angle"] = fs.get()[:2]
    # bounded map: points falling outside are dropped (has_cell)
    mb = R.map_create(REF_CELL_MEAN, MAP_PLAIN, 60, 60, 0.1)
    spe_b = R.spe_create(OOPE_OBSTACLE, OIE_DISCREPANCY, 0)
    fsb = R.filter_scan(spe_b, scan2, (0.3, -0.2, 0.4), mb)
    out["filt_bounded_range"], out["filt_bounded_angle"] = fsb.get()[:2]
    save("weights_ahr.npz", **out)


def gen_world_to_cells(R):
    rs = np.random.RandomState(3)
    m = R.map_create(REF_CELL_MEAN, MAP_UNBOUNDED_PLAIN, 100, 100, 0.1)
    segs = list(rs.uniform(-3, 3, size=(60, 4)))
    segs += [[0.05, 0.05, 2.05, 0.05], [0.05, 0.05, 0.05, -1.95], [0.05, 0.05, 1.05, 1.05],
             [0.0, 0.0, 1.0, 1.0], [0.0, 0.0, -1.0, 1.0], [0.1, 0.1, 0.1, 0.9], [0.25, 0.25, 0.25, 0.25],
             [0.05, 0.05, -1.95, -0.95], [1.0, 0.05, -1.0, 0.05], [0.3, 0.3, 0.7, 0.5]]
    segs = np.array(segs)
    cells, offs = [], [0]
    for s in segs:
        c = R.world_to_cells(m, *s)
        cells.append(c)
        offs.append(offs[-1] + len(c))
    save("world_to_cells.npz", scale=np.array(0.1), segments=segs, cells=np.concatenate(cells),
         offsets=np.array(offs))


def main():
    if not ref_available():
        sys.exit("oracle/_ref/libslamref.so missing: run `make -C oracle ref` where /root/reference exists")
    R = Ref()
    gen_enumerators(R)
    gen_oope_known(R)
    tbm_base = (0.95, 0.04, 0.01, 0.003)   # config/slams/viny_slam_base.properties:10-13
    std_base = (0.95, 1.0, 0.01, 1.0)      # config/slams/tiny_slam_base.properties:9-11
    for trig, tn in [(TRIG_RAW, "raw"), (TRIG_CACHED, "cached")]:
        gen_scene(R, REF_CELL_MEAN, "mean", 0, "even", trig, tn, std_base)
        gen_scene(R, REF_CELL_TBM, "tbm", 1, "viny", trig, tn, tbm_base)
    gen_scene(R, REF_CELL_AFFINE, "affine", 0, "even", TRIG_RAW, "raw", std_base)
    gen_hc_smoke(R)
    gen_gmapping_scene(R)
    gen_gmapping_pf(R)
    gen_resample(R)
    gen_weights_ahr(R)
    gen_world_to_cells(R)


if __name__ == "__main__":
    main()
