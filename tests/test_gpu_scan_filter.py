"""GPU suite: slamhip_scan_filter_upload -- the raw scan in, `filter_scan` + weighting + beam trig + upload in one call
(weighted_mean_point_probability_spe.h:75-95, :21-60; trigonometry_utils.h:17-84) -- against the same steps taken one
by one through the entry points that are pinned to the reference goldens (slamhip_filter_scan, slamhip_scan_weights,
slamhip_beam_trig_*, slamhip_scan_upload): kept indices equal, scores of a pose cloud bit for bit."""
import numpy as np
import pytest

import __graft_entry__ as ge
from synth import make_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    return ge.load_package()


@pytest.mark.parametrize("weighting", ["even", "viny", "ahr"])
@pytest.mark.parametrize("trig", ["raw", "cached"])
@pytest.mark.parametrize("bounded", [False, True])
def test_scan_filter_upload_equals_the_separate_steps(pkg, weighting, trig, bounded):
    sc = make_scene(cell_model=0, size=600, scale=0.05, n_beams=720, seed=21, weighting="even")
    m = sc["map"]
    ctx = pkg.Context(0)
    ctx.upload_map(0, m)
    rs = np.random.RandomState(5)
    n = 720
    inc = np.deg2rad(270.0) / n
    a_min, a_max = -np.deg2rad(135.0), -np.deg2rad(135.0) + inc * n + inc
    ang = a_min + inc * np.arange(n)
    if trig == "cached":  # the provider's own angles (the accumulating loop), so that every index is exact
        acc, a = [], a_min
        while a < a_max:
            acc.append(a)
            a += inc
        ang = np.array(acc[:n])
    geom = dict(width=m.width, height=m.height, origin=m.origin, scale=m.scale, bounded=bounded)
    cfg = pkg.spe_cfg(sum_order=pkg.SUM_SEQUENTIAL, pose_trig=pkg.POSE_TRIG_HOST)
    poses = sc["true_pose"] + rs.randn(32, 3) * [0.1, 0.1, 0.05]
    for rep, (skip, max_range) in enumerate([(0, -1.0), (3, -1.0), (0, 9.0), (2, 14.0), (0, -1.0)]):
        rng = 1.0 + 17.0 * rs.rand(n)  # long beams leave the 30 m window: a bounded map drops them
        occ = (rs.rand(n) < 0.85).astype(np.int32)
        fac = 0.5 + rs.rand(n)
        pose = sc["true_pose"] + rs.randn(3) * [0.2, 0.2, 0.1]
        if rep == 4:
            ang = ang + 1e-3 if trig == "raw" else ang  # the cached per-angle quantities must notice a new angle array
        tm = pkg.TRIG_CACHED if trig == "cached" else pkg.TRIG_RAW
        # one by one
        tab = pkg.beam_trig(ang, tm, a_min, a_max, inc)
        kept = pkg.filter_scan(rng, ang, occ, pose, geom, skip_rate=skip, max_range=max_range, trig_mode=tm, a_min=a_min,
                               a_delta=inc, tab_sin=tab[1] if trig == "cached" else None,
                               tab_cos=tab[0] if trig == "cached" else None)
        assert 0 < kept.size < n
        w = pkg.scan_weights(weighting, rng[kept], ang[kept])
        ctx.scan_upload(rng[kept], tab[0][kept], tab[1][kept], w, fac[kept])
        want = ctx.score_poses(0, cfg, poses)
        # in one call
        got_kept = ctx.scan_filter_upload(0, rng, ang, pose, is_occ=occ, factor=fac, trig_mode=tm, a_min=a_min, a_max=a_max,
                                          a_inc=inc, skip_rate=skip, max_range=max_range, bounded=bounded, weighting=weighting)
        np.testing.assert_array_equal(got_kept, kept, err_msg="rep %d" % rep)
        got = ctx.score_poses(0, cfg, poses)
        np.testing.assert_array_equal(got, want, err_msg="rep %d" % rep)
    ctx.close()


def test_process_raw_scan_is_filter_upload_plus_process_scan(pkg):
    """r06: slamhip_matcher_process_raw_scan -- the reference's process_scan signature (raw scan and initial pose in, pose
    delta and probability out; grid_scan_matcher.h:153-156) -- against slamhip_scan_filter_upload + slamhip_matcher_process_scan
    on the same raw scans: same points kept, same result bit for bit, for the three matcher kinds; a scan of which
    filter_scan keeps nothing comes back as the reference's NaN with a zero delta."""
    from synth import cast_scan
    ctx = pkg.Context(0)
    try:
        sc = make_scene(cell_model=0, size=600, scale=0.05, n_beams=720, seed=21)
        ctx.upload_map(0, sc["map"])
        rng, ang, occ = cast_scan(sc["gt"], sc["map"].scale, sc["true_pose"], 720, seed=5, raw=True)
        for kind, prm in (("HC", [20, 0.1, 0.1]), ("MC", [7, 0.2, 0.1, 20, 100]),
                          ("BF", [-0.1, 0.1, 0.05, -0.1, 0.1, 0.05, -0.05, 0.05, 0.025])):
            two, one = pkg.Matcher(ctx, kind, pkg.spe_cfg(), prm), pkg.Matcher(ctx, kind, pkg.spe_cfg(), prm)
            up = ctx.make_raw_scan(0, rng, ang, is_occ=occ, skip_rate=2, max_range=20.0)
            match = one.make_raw_process_scan(0, rng, ang, is_occ=occ, skip_rate=2, max_range=20.0)
            for k in range(3):
                pose = sc["init_pose"] + k * np.array([0.01, -0.02, 0.005])
                kept2 = up(pose)
                r2 = two.process_scan(0, pose)
                kept1, prob1 = match(pose)
                assert kept1 == kept2 and 0 < kept1 < 720
                assert prob1 == r2["prob"] and np.array_equal(np.array(list(match.delta)), r2["delta"])
                assert one.stats()["scorer_calls"] == two.stats()["scorer_calls"]
            empty = one.make_raw_process_scan(0, rng, ang, is_occ=np.zeros_like(occ))
            kept, prob = empty(sc["init_pose"])
            assert kept == 0 and np.isnan(prob) and list(empty.delta) == [0.0, 0.0, 0.0]
            two.close()
            one.close()
    finally:
        ctx.close()
