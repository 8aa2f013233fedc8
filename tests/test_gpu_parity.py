"""GPU suite (-m gpu): the HIP path, called through the C-ABI (libslamhip.so), against
 (1) the golden vectors captured from the compiled reference (tests/golden/),
 (2) the CPU oracle on the same seeded inputs at BASELINE.json sizes,
 (3) size-independent properties (batch-shape independence, tie consistency, permutation).

Parity bars: per-pose scores within 1e-5 relative is the contract of north_star; what
these tests actually demand is much tighter --
  * SLAMHIP_SUM_SEQUENTIAL + host pose trig : bit-exact scores, traces, deltas
  * default (canonical tree sum, device sincos): scores within 1e-12 relative, identical accept
    traces / candidate poses / deltas, and bit-exact equality with the oracle's tree-order mode
  * GMapping OOPE (device exp): 1e-11 relative
"""
import numpy as np
import pytest
from helpers import SCENES, assert_trace_equal, filtered_scan, load, map_from, trace

import __graft_entry__ as ge

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    return ge.load_package()


@pytest.fixture(scope="module")
def ctx(pkg):
    c = pkg.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def po():
    import pyoracle
    return pyoracle


def upload_scene(pkg, ctx, m, scan, map_id=0):
    ctx.upload_map(map_id, m)
    if getattr(scan, "trig_mode", 0) == 1:
        idx = np.round((scan.angle - scan.a_min) / scan.a_delta).astype(np.int64)
        cos_a, sin_a = scan.tab_cos[idx], scan.tab_sin[idx]
    else:
        cos_a, sin_a = pkg.beam_trig(scan.angle)
    ctx.scan_upload(scan.range, cos_a, sin_a, scan.weight, scan.factor)


STRICT = dict(sum_order=1, pose_trig=1)  # SLAMHIP_SUM_SEQUENTIAL, SLAMHIP_POSE_TRIG_HOST


@pytest.mark.parametrize("scene", SCENES)
def test_scores_vs_reference_golden(pkg, ctx, po, oracle, scene):
    g = load("scene_%s.npz" % scene)
    m, scan = map_from(g), filtered_scan(g)
    upload_scene(pkg, ctx, m, scan)
    strict = ctx.score_poses(0, pkg.spe_cfg(**STRICT), g["poses"])
    np.testing.assert_array_equal(strict, g["scores"])  # bit-exact with the reference
    tree_host = ctx.score_poses(0, pkg.spe_cfg(pose_trig=1), g["poses"])
    want_tree = oracle.score_poses(m, scan, po.make_cfg(sum_order=po.SUM_TREE256), g["poses"])
    np.testing.assert_array_equal(tree_host, want_tree)  # canonical order restated on the CPU
    dflt = ctx.score_poses(0, pkg.spe_cfg(), g["poses"])
    np.testing.assert_allclose(dflt, g["scores"], rtol=1e-12, atol=0)
    assert dflt[0] == dflt[1] and strict[0] == strict[1]  # identical poses -> identical bits


@pytest.mark.parametrize("scene", SCENES)
@pytest.mark.parametrize("matcher", ["mc", "mc_long", "hc6", "hc128"])
def test_matcher_traces_vs_reference_golden(pkg, ctx, scene, matcher):
    g = load("scene_%s.npz" % scene)
    m, scan = map_from(g), filtered_scan(g)
    upload_scene(pkg, ctx, m, scan)
    kind = {0: "MC", 1: "HC"}[int(g[matcher + "_kind"])]
    ref = trace(g, matcher + "_")
    mt = pkg.Matcher(ctx, kind, pkg.spe_cfg(**STRICT), g[matcher + "_params"])
    t = mt.process_scan(0, g["init_pose"], trace=True)
    assert_trace_equal(t, ref)  # bit-exact: poses, scores, accept flags, delta, prob
    st = mt.stats()
    assert st["scorer_calls"] == ref["n_calls"] and st["poses_evaluated"] >= ref["n_calls"]
    if matcher == "mc":  # the engine is not reseeded between process_scan calls (Q7)
        t2 = mt.process_scan(0, g["init_pose"], trace=True)
        assert_trace_equal(t2, trace(g, "mc_second_"))
    # default mode: same decisions, scores to 1e-12
    md = pkg.Matcher(ctx, kind, pkg.spe_cfg(), g[matcher + "_params"])
    td = md.process_scan(0, g["init_pose"], trace=True)
    assert_trace_equal(td, ref, exact_scores=False, rtol=1e-12)
    if matcher == "mc":  # ... and the device chain leaves the engine where the reference's loop does
        assert_trace_equal(md.process_scan(0, g["init_pose"], trace=True), trace(g, "mc_second_"), exact_scores=False, rtol=1e-12)
    # speculation depth must not change the result
    for batch in (1, 7, 64):
        mb = pkg.Matcher(ctx, kind, pkg.spe_cfg(**STRICT), g[matcher + "_params"])
        mb.set_batch(batch)
        assert_trace_equal(mb.process_scan(0, g["init_pose"], trace=True), ref)


def test_hc_smoke_cases_of_the_reference(pkg, ctx, po, oracle):
    """hill_climbing_sm_smoke_test.cpp:72-105 through the HIP matcher.

    The fixture geometry puts endpoints EXACTLY on cell boundaries (robot on a cell centre, steps
    of half a cell), so with the raw trig provider the last ulp of libm's sin(theta + a) decides
    the cell.  The device evaluates the angle-addition form (= the reference's own
    CachedTrigonometryProvider arithmetic), hence:
      * cached provider: the whole trace is bit-exact with the reference;
      * raw provider: the reference test's OWN acceptance rule is applied (recovered pose exact
        or equal scan probability, scan_matcher_test_utils.h:46-80)."""
    g = load("hc_smoke.npz")
    m = map_from(g)
    ctx.upload_map(0, m)
    geom = dict(width=m.width, height=m.height, origin=m.origin, scale=m.scale, bounded=False)
    a_min, a_inc = float(g["a_min"]), float(g["a_inc"])
    for i, nz in enumerate(g["noises"]):
        pose = g["rpose"] + nz
        # -- cached provider: exact trace
        kept = pkg.filter_scan(g["raw_range"], g["raw_angle"], g["raw_occ"], pose, geom,
                               trig_mode=pkg.TRIG_CACHED, a_min=a_min, a_delta=a_inc,
                               tab_sin=g["tab_sin"], tab_cos=g["tab_cos"])
        r, a = g["raw_range"][kept], g["raw_angle"][kept]
        c, s = pkg.beam_trig(a, pkg.TRIG_CACHED, a_min, float(g["a_max_passed"]), a_inc)
        ctx.scan_upload(r, c, s, pkg.scan_weights("even", r, a))
        mt = pkg.Matcher(ctx, "HC", pkg.spe_cfg(**STRICT), g["params"])
        assert_trace_equal(mt.process_scan(0, pose, trace=True), trace(g, "cached%d_" % i))
        # -- raw provider: the reference test's acceptance rule
        kept = pkg.filter_scan(g["raw_range"], g["raw_angle"], g["raw_occ"], pose, geom)
        r, a = g["raw_range"][kept], g["raw_angle"][kept]
        c, s = pkg.beam_trig(a)
        ctx.scan_upload(r, c, s, pkg.scan_weights("even", r, a))
        mt = pkg.Matcher(ctx, "HC", pkg.spe_cfg(**STRICT), g["params"])
        t = mt.process_scan(0, pose, trace=True)
        result_noise = nz + t["delta"]
        k0 = pkg.filter_scan(g["raw_range"], g["raw_angle"], g["raw_occ"], g["rpose"], geom)
        r0, a0 = g["raw_range"][k0], g["raw_angle"][k0]
        c0, s0 = pkg.beam_trig(a0)
        ctx.scan_upload(r0, c0, s0, pkg.scan_weights("even", r0, a0))
        p_true, p_res = ctx.score_poses(0, pkg.spe_cfg(**STRICT), np.stack([g["rpose"], g["rpose"] + result_noise]))
        same_prob = abs(p_true - p_res) <= 1e-7 * max(1.0, abs(p_true), abs(p_res))
        assert same_prob or np.all(np.abs(result_noise) <= np.finfo(np.float64).eps), (i, result_noise)


def test_gmapping_scores_and_trace_vs_reference_golden(pkg, ctx, po, oracle):
    g = load("gmapping_scene.npz")
    m, scan = map_from(g), filtered_scan(g)
    upload_scene(pkg, ctx, m, scan)
    cfg = pkg.spe_cfg(oope=pkg.OOPE_GMAPPING, pose_trig=1)
    ctx.gm_cache_reset()
    s = ctx.score_poses(0, cfg, g["poses"])  # ONE call sequence: the cache carries across poses
    np.testing.assert_allclose(s, g["scores"], rtol=1e-11, atol=1e-300)
    # cache state after the sequence equals the oracle's
    cache = po.Oracle.new_gm_cache()
    oracle.score_poses(m, scan, po.make_cfg(oope=po.OOPE_GMAPPING), g["poses"], cache)
    cx, cy, pr = ctx.gm_cache_get()
    assert (cx, cy) == (cache.cx, cache.cy) and abs(pr - cache.prob) <= 1e-11 * max(pr, 1e-300)
    # split into two calls: same answers (carry survives the call boundary)
    ctx.gm_cache_reset()
    s2 = np.concatenate([ctx.score_poses(0, cfg, g["poses"][:17]), ctx.score_poses(0, cfg, g["poses"][17:])])
    np.testing.assert_array_equal(s, s2)
    # HC(6, 0.1, 0.1) with sp_skip_rate = 3 (launch/gmapping_mit_run.launch)
    r3, a3 = g["skip3_range"], g["skip3_angle"]
    c, sn = pkg.beam_trig(a3)
    ctx.scan_upload(r3, c, sn, pkg.scan_weights("even", r3, a3))
    ctx.gm_cache_reset()
    mt = pkg.Matcher(ctx, "HC", cfg, [6, 0.1, 0.1])
    t = mt.process_scan(0, g["init_pose"], trace=True)
    assert_trace_equal(t, trace(g, "hc6_skip3_"), exact_scores=False, rtol=1e-11)


@pytest.mark.parametrize("cell,weighting,size,scale,beams", [
    (0, "even", 2000, 0.05, 1080),   # BASELINE cfg 2 shape (tinySLAM, occupancy cell)
    (1, "viny", 2000, 0.05, 1080),   # cfg 3 shape (vinySLAM, TBM cell)
    (0, "even", 1000, 0.1, 720),     # cfg 1 shape
])
def test_full_size_against_oracle_and_properties(pkg, ctx, po, oracle, cell, weighting, size, scale, beams):
    from synth import make_scene
    sc = make_scene(cell_model=cell, size=size, scale=scale, n_beams=beams, seed=5, weighting=weighting)
    m, scan = sc["map"], sc["scan"]
    upload_scene(pkg, ctx, m, scan)
    rs = np.random.RandomState(9)
    P = 4096
    poses = sc["init_pose"] + rs.randn(P, 3) * [0.2, 0.2, 0.1]
    poses[0] = sc["init_pose"]
    strict = ctx.score_poses(0, pkg.spe_cfg(**STRICT), poses)
    tree = ctx.score_poses(0, pkg.spe_cfg(pose_trig=1), poses)
    dflt = ctx.score_poses(0, pkg.spe_cfg(), poses)
    sub = rs.choice(P, 256, replace=False)
    want = oracle.score_poses(m, scan, po.make_cfg(), poses[sub])
    np.testing.assert_array_equal(strict[sub], want)
    want_tree = oracle.score_poses(m, scan, po.make_cfg(sum_order=po.SUM_TREE256), poses[sub])
    np.testing.assert_array_equal(tree[sub], want_tree)
    np.testing.assert_allclose(dflt, strict, rtol=1e-12, atol=0)
    # batch-shape independence: any split of the batch gives the same bits (canonical order)
    for cfgk in (dict(), STRICT):
        full = ctx.score_poses(0, pkg.spe_cfg(**cfgk), poses)
        parts = np.concatenate([ctx.score_poses(0, pkg.spe_cfg(**cfgk), poses[a:b])
                                for a, b in [(0, 1), (1, 8), (8, 1000), (1000, 3333), (3333, P)]])
        np.testing.assert_array_equal(full, parts)
    # permutation of the batch permutes the scores
    perm = rs.permutation(P)
    np.testing.assert_array_equal(ctx.score_poses(0, pkg.spe_cfg(), poses[perm]), dflt[perm])
    # matcher at full size: exact trace vs the oracle's accept loop
    # (the last entry is BASELINE configs[2]'s own matcher: 4096 attempts, 4097 scorer calls)
    for kind, okind, prm in (("HC", po.SM_HC, [128, 0.1, 0.1]), ("MC", po.SM_MC, [666666, 0.2, 0.1, 64, 512]),
                             ("MC", po.SM_MC, [666666, 0.2, 0.1, 4096, 4096])):
        mt = pkg.Matcher(ctx, kind, pkg.spe_cfg(**STRICT), prm)
        t = mt.process_scan(0, sc["init_pose"], trace=True)
        e = oracle.enumerator(okind, prm)
        r = oracle.process_scan(e, m, scan, po.make_cfg(), sc["init_pose"])
        assert_trace_equal(t, r)
        td = pkg.Matcher(ctx, kind, pkg.spe_cfg(), prm).process_scan(0, sc["init_pose"], trace=True)
        assert_trace_equal(td, r, exact_scores=False, rtol=1e-12)


def test_edge_cases(pkg, ctx, po, oracle):
    from synth import MapData, Scan
    rs = np.random.RandomState(2)
    pay = rs.rand(37, 53)
    m = MapData(0, pay[:, :, None], (20, 11), 0.25, [0.5])
    for n in (1, 2, 63, 64, 65, 255, 256, 257, 1081, 1300, 2500):  # KB=1..5 and the generic path
        r = rs.uniform(0.2, 6.0, n)
        a = np.sort(rs.uniform(-2.3, 2.3, n))
        w = rs.rand(n) + 0.1
        f = np.where(rs.rand(n) < 0.2, rs.rand(n), 1.0)
        scan = Scan(r, a, w, f)
        upload_scene(pkg, ctx, m, scan)
        poses = rs.randn(33, 3) * [2.0, 2.0, 1.5]
        poses[0] = [1e3, -1e3, 0.1]  # everything outside the window -> unknown cell
        got = ctx.score_poses(0, pkg.spe_cfg(**STRICT), poses)
        want = oracle.score_poses(m, scan, po.make_cfg(), poses)
        np.testing.assert_array_equal(got, want)
        occ = ctx.score_poses(0, pkg.spe_cfg(oie=pkg.OIE_OCCUPANCY, **STRICT), poses)
        want_occ = oracle.score_poses(m, scan, po.make_cfg(oie=po.OIE_OCCUPANCY), poses)
        np.testing.assert_array_equal(occ, want_occ)
    # zero total weight -> quiet NaN (weighted_mean_point_probability_spe.h:127-131)
    scan = Scan([1.0, 2.0], [0.0, 0.1], [0.0, 0.0])
    upload_scene(pkg, ctx, m, scan)
    assert np.all(np.isnan(ctx.score_poses(0, pkg.spe_cfg(), np.zeros((3, 3)))))
    # endpoints exactly on cell boundaries: floor(x / scale) with a true division
    scan = Scan([0.25, 0.5, 0.75, 1.0], [0.0, 0.0, np.pi / 2, np.pi], np.full(4, 0.25))
    upload_scene(pkg, ctx, m, scan)
    poses = np.array([[0.0, 0.0, 0.0], [0.25, -0.25, 0.0], [-0.75, 0.5, 0.0]])
    np.testing.assert_array_equal(ctx.score_poses(0, pkg.spe_cfg(**STRICT), poses),
                                  oracle.score_poses(m, scan, po.make_cfg(), poses))


def test_map_mirror_dirty_log_and_growth(pkg, ctx, po, oracle):
    from synth import MapData, Scan
    rs = np.random.RandomState(4)
    pay = rs.rand(40, 48, 4)
    pay /= pay.sum(axis=2, keepdims=True)
    m = MapData(1, pay, (24, 20), 0.1, [1.0, 0.0, 0.0, 0.0])
    scan = Scan(rs.uniform(0.3, 2.0, 300), np.sort(rs.uniform(-2, 2, 300)), np.full(300, 1 / 300))
    upload_scene(pkg, ctx, m, scan, map_id=3)
    poses = rs.randn(40, 3) * [0.5, 0.5, 1.0]
    np.testing.assert_array_equal(ctx.score_poses(3, pkg.spe_cfg(**STRICT), poses),
                                  oracle.score_poses(m, scan, po.make_cfg(), poses))
    # GridMap::update forwarded as a dirty log
    xy = np.stack([rs.randint(0, 48, 200), rs.randint(0, 40, 200)], axis=1)
    _, first = np.unique(xy[:, 1] * 48 + xy[:, 0], return_index=True)
    xy = xy[np.sort(first)]
    vals = rs.rand(len(xy), 4)
    m.payload[xy[:, 1], xy[:, 0]] = vals
    ctx.map_apply_dirty(3, xy, vals)
    np.testing.assert_array_equal(ctx.map_download_window(3, 0, 0, 48, 40, 4), m.payload)
    np.testing.assert_array_equal(ctx.score_poses(3, pkg.spe_cfg(**STRICT), poses),
                                  oracle.score_poses(m, scan, po.make_cfg(), poses))
    # growth of an unbounded map: more cells on every side, origin shifts, old cells keep their
    # external coordinates (plain_grid_map.h:133-173)
    big = np.tile(np.array([1.0, 0.0, 0.0, 0.0]), (60, 70, 1))
    big[7:47, 9:57] = m.payload
    m2 = MapData(1, big, (24 + 9, 20 + 7), 0.1, [1.0, 0.0, 0.0, 0.0])
    ctx.map_bind(3, 1, 70, 60, m2.origin, 0.1, m2.unknown)
    np.testing.assert_array_equal(ctx.map_download_window(3, 0, 0, 70, 60, 4), big)
    np.testing.assert_array_equal(ctx.score_poses(3, pkg.spe_cfg(**STRICT), poses),
                                  oracle.score_poses(m2, scan, po.make_cfg(), poses))
    ctx.map_release(3)
    with pytest.raises(pkg.SlamHipError):
        ctx.score_poses(3, pkg.spe_cfg(), poses)


def test_smoke_entry(pkg):
    ge.smoke()


@pytest.mark.parametrize("pose_trig", [1, 0])
@pytest.mark.parametrize("scenario", ["default", "nogate", "wide"])
def test_gmapping_filter_vs_reference_golden(pkg, ctx, scenario, pose_trig):
    """G5 through the C-ABI: several GmappingParticleFilter::handle_sensor_data steps (gate, pose
    noise, HC matching of every particle with the shared OOPE cache, weights, N_eff, resampling with
    duplicated particles, master hand-over) against the compiled reference.  pose_trig 1: host trigonometry,
    host-driven lock-step jobs; 0 (the default): one device chain per particle in shared launches."""
    g = load("gmapping_pf.npz")
    m = map_from(g, scenario + "_map_")
    ctx.upload_map(5, m)
    n = len(g[scenario + "_seeds"])
    pf = pkg.GmappingFilter(ctx, pkg.gmapping_params(gp8=g[scenario + "_gp"], skip_rate=3, pose_trig=pose_trig),
                            n, g[scenario + "_seeds"])
    resampled_any = False
    for k in range(int(g[scenario + "_n_steps"])):
        pre = "%s_step%d_" % (scenario, k)
        res, _idx = pf.step(5, g[pre + "range"], g[pre + "angle"], None, g[pre + "delta"], 7 + k)
        poses, w, ms = pf.state()
        assert res == bool(int(g[pre + "resampled"])), k
        np.testing.assert_array_equal(ms, g[pre + "master"])
        np.testing.assert_allclose(poses, g[pre + "poses"], rtol=0, atol=1e-10)
        np.testing.assert_allclose(w, g[pre + "weights"], rtol=1e-9, atol=0)
        resampled_any |= res
    assert resampled_any == (scenario == "wide")
    ctx.map_release(5)


@pytest.mark.parametrize("pose_trig", [1, 0])
@pytest.mark.parametrize("size", [2000, 4000])
def test_gmapping_filter_100_particles_vs_oracle(pkg, ctx, po, oracle, size, pose_trig):
    """BASELINE cfg 4 (100 particles, 1080 beams, GMapping cell/OOPE, HC(6, 0.1, 0.1)) on a 2000x2000 and on the
    configuration's own 4000x4000 @0.05 m map (0.5 GB of cells in HBM) against the sequential CPU oracle, both ways
    the filter can run its likelihood step: pose_trig 1 = host pose trigonometry, host-driven lock-step jobs;
    pose_trig 0 = the DEFAULT and what bench.py times (`pf` leg): device sincos, one accept chain per particle on the
    device in shared launches (csrc/hc_chain.hip, grid.y = particle).  Same bars either way: resampling indices and
    scorer calls exact, poses 1e-10, weights 1e-9 (gmapping_world.h:73-101)."""
    from synth import make_scene
    sc = make_scene(cell_model=2, size=size, scale=0.05, n_beams=1080, seed=11)
    m, scan = sc["map"], sc["scan"]
    ctx.upload_map(6, m)
    n = 100
    seeds = np.arange(1000, 1000 + n, dtype=np.uint32)
    gp = [0.0, 0.1, 0.0, 0.03, 0.0, 0.0, 0.0, 0.0]  # gate open: every particle matches
    pf = pkg.GmappingFilter(ctx, pkg.gmapping_params(gp8=gp, pose_trig=pose_trig), n, seeds)
    opf = oracle.gmapping_create(n, gp, seeds)
    deltas = [sc["true_pose"], [0.02, 0.01, 0.01], [0.4, 0.5, 0.3], [0.01, -0.02, 0.02]]
    for k, d in enumerate(deltas):
        res, idx = pf.step(6, scan.range, scan.angle, None, d, 7 + k)
        ores, oidx = opf.step(m, scan.range, scan.angle, None, d, 7 + k)
        poses, w, ms = pf.state()
        oposes, ow, oms = opf.state()
        assert res == ores
        if res:
            np.testing.assert_array_equal(idx, oidx)  # resampling indices bit-exact
        np.testing.assert_array_equal(ms, oms)
        np.testing.assert_allclose(poses, oposes, rtol=0, atol=1e-10)
        np.testing.assert_allclose(w, ow, rtol=1e-9, atol=0)
        st = pf.stats()
        assert st["scorer_calls"] == opf.o.lib.orc_gmapping_scorer_calls(opf.h)
    ctx.map_release(6)


def test_window_oopes_vs_reference(pkg, ctx, po, oracle):
    """K2: max / mean / overlap OOPEs -- the 18 known-answer cases of the reference
    (occupancy_observation_probability_test.cpp:59-207) as one-beam scans, and the scene goldens."""
    from synth import Scan
    g = load("oope_known.npz")
    m = map_from(g)
    ctx.upload_map(0, m)
    one = Scan([0.0], [0.0], [1.0])
    upload_scene(pkg, ctx, m, one)
    eps = np.finfo(np.float64).eps
    for kind, lit, ref_out, obst, r in zip(g["kinds"], g["expected_literal"], g["reference_out"],
                                          g["obstacle"], g["range4"]):
        cfg = pkg.spe_cfg(oope=int(kind), area=r, **STRICT)
        got = ctx.score_poses(0, cfg, [[obst[0], obst[1], 0.0]])[0]
        assert got == ref_out and abs(got - lit) <= eps, (kind, obst, r, got, ref_out)
    for scene in ("mean_raw", "tbm_cached"):
        g = load("scene_%s.npz" % scene)
        m, scan = map_from(g), filtered_scan(g)
        upload_scene(pkg, ctx, m, scan)
        for name, kind in (("max", pkg.OOPE_MAX), ("mean", pkg.OOPE_MEAN), ("overlap", pkg.OOPE_OVERLAP)):
            got = ctx.score_poses(0, pkg.spe_cfg(oope=kind, area=g["win_area"], **STRICT), g["poses"][:24])
            if name == "overlap" and scene.endswith("raw"):
                # overlap weights depend continuously on the endpoint: with the raw trig provider
                # the reference's libm sin(theta + a) and the device's angle-addition form differ
                # in the last ulp; exact with the cached provider
                np.testing.assert_allclose(got, g["win_%s_scores" % name], rtol=1e-13, atol=0)
            else:
                np.testing.assert_array_equal(got, g["win_%s_scores" % name], err_msg=name + scene)
            dflt = ctx.score_poses(0, pkg.spe_cfg(oope=kind, area=g["win_area"]), g["poses"][:24])
            np.testing.assert_allclose(dflt, g["win_%s_scores" % name], rtol=1e-12, atol=0)


def test_brute_force_matcher_vs_oracle(pkg, ctx, po, oracle):
    """N1: BruteForceScanMatcher (brute_force_scan_matcher.h:10-81) -- the embarrassingly parallel
    caller of the same scorer; p2D_ss_evaluator-style sweep."""
    g = load("scene_mean_raw.npz")
    m, scan = map_from(g), filtered_scan(g)
    upload_scene(pkg, ctx, m, scan)
    rng9 = [-0.3, 0.3, 0.05, -0.2, 0.2, 0.05, -0.06, 0.06, 0.02]
    mt = pkg.Matcher(ctx, "BF", pkg.spe_cfg(**STRICT), rng9)
    t = mt.process_scan(0, g["init_pose"], trace=True)
    e = oracle.enumerator(po.SM_BF, rng9)
    r = oracle.process_scan(e, m, scan, po.make_cfg(), g["init_pose"])
    assert_trace_equal(t, r)
    assert mt.stats()["launches"] <= 2 and t["n_calls"] > 500


@pytest.mark.parametrize("scale", [0.05, 0.1, 0.025, 1.0 / 3.0, 0.07])
def test_cell_index_on_boundaries_matches_true_division(pkg, ctx, po, oracle, scale):
    """to_cell() multiplies by 1/scale and falls back to the true quotient near integers: every
    cell of the map holds a different value, endpoints sit exactly on (or one ulp off) cell
    boundaries, so one wrong floor(x / scale) changes a score."""
    from synth import MapData, Scan
    rs = np.random.RandomState(int(scale * 1000))
    pay = rs.rand(96, 96)
    m = MapData(0, pay[:, :, None], (48, 48), scale, [0.5])
    k = rs.randint(1, 30, 512)
    r = k * scale  # ranges that are whole numbers of cells
    a = rs.choice([0.0, np.pi / 2, np.pi, -np.pi / 2, np.pi / 4], 512)
    scan = Scan(r, a, np.full(512, 1 / 512))
    upload_scene(pkg, ctx, m, scan)
    base = rs.randint(-10, 10, (256, 2)) * scale
    poses = np.zeros((256 * 3, 3))
    poses[:256, :2] = base
    poses[256:512, :2] = np.nextafter(base, np.inf)
    poses[512:, :2] = np.nextafter(base, -np.inf)
    poses[:, 2] = rs.choice([0.0, np.pi / 2, -np.pi / 2], 768)
    got = ctx.score_poses(0, pkg.spe_cfg(**STRICT), poses)
    want = oracle.score_poses(m, scan, po.make_cfg(), poses)
    np.testing.assert_array_equal(got, want)


def _pf_twin(pkg, chains):
    from synth import make_scene
    ctx = pkg.Context(0)
    ctx.set_option(pkg.OPT_FILTER_CHAINS, 1 if chains else 0)
    sc = make_scene(cell_model=2, size=1500, scale=0.05, n_beams=1080, seed=13)
    ctx.upload_map(0, sc["map"])
    out = {}
    for n in (100, 13, 1):
        gp = [0.0, 0.1, 0.0, 0.03, 0.0, 0.0, 0.0, 0.0]
        pf = pkg.GmappingFilter(ctx, pkg.gmapping_params(gp8=gp), n, np.arange(500, 500 + n, dtype=np.uint32))
        for k, d in enumerate([sc["true_pose"], [0.02, 0.01, 0.01], [0.3, 0.2, 0.2], [0.01, -0.02, 0.02], [0.0, 0.05, -0.01]]):
            res, idx = pf.step(0, sc["scan"].range, sc["scan"].angle, None, d, 7 + k)
            poses, w, ms = pf.state()
            out["n%d_s%d_poses" % (n, k)], out["n%d_s%d_w" % (n, k)] = poses, w
            out["n%d_s%d_calls" % (n, k)] = np.array([pf.stats()["scorer_calls"], int(res)])
        pf.close()
    ctx.close()
    return out


def test_filter_chains_on_the_device_equal_the_lock_step_jobs(pkg):
    """The likelihood-only filter step runs one hill-climbing chain per particle on the device, all chains in shared
    launches; SLAMHIP_OPT_FILTER_CHAINS = 0 keeps the host-driven lock-step jobs.  100 / 13 / 1
    particles, five steps each with resamplings -- poses, weights, scorer calls and resampling decisions bit for bit."""
    a, b = _pf_twin(pkg, True), _pf_twin(pkg, False)
    assert set(a) == set(b) and len(a) == 3 * 5 * 3
    for k in a:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)


def _pf_maps_twin(pkg, chains):
    from synth import make_scene
    ctx = pkg.Context(0)
    ctx.set_option(pkg.OPT_FILTER_CHAINS, 1 if chains else 0)
    size = 1500
    sc = make_scene(cell_model=2, size=size, scale=0.05, n_beams=1080, seed=13)
    m = sc["map"]
    ctx.upload_map(0, m)
    out = {}
    for n in (40, 5):
        gp = [0.0, 0.1, 0.0, 0.03, 0.0, 0.0, 0.0, 0.0]
        pf = pkg.GmappingFilter(ctx, pkg.gmapping_params(gp8=gp), n, np.arange(500, 500 + n, dtype=np.uint32))
        pf.enable_particle_maps(0, extent_tiles=(size + 127) // 128 + 1, pool_tiles=200 + 120 * n)
        for k, d in enumerate([sc["true_pose"], [0.02, 0.01, 0.01], [0.3, 0.2, 0.2], [0.01, -0.02, 0.02]]):
            res, idx = pf.step(0, sc["scan"].range, sc["scan"].angle, None, d, 7 + k)
            poses, w, ms = pf.state()
            out["n%d_s%d_poses" % (n, k)], out["n%d_s%d_w" % (n, k)] = poses, w
            out["n%d_s%d_calls" % (n, k)] = np.array([pf.stats()["scorer_calls"], int(res)])
        ox, oy = m.origin
        for i in (0, n - 1):
            pay, aux = pf.particle_map(i, -ox, -oy, size, size)
            out["n%d_map%d" % (n, i)], out["n%d_aux%d" % (n, i)] = pay, aux
        pf.close()
    ctx.close()
    return out


def test_filter_chains_through_tile_tables_equal_the_lock_step_jobs(pkg):
    """The same twin run with per-particle copy-on-write maps: every chain gathers through the tile table of its own
    particle (HcChainArgs::tables / slots), and the step ends with the batched map update.  40 and 5 particles, four
    steps with resamplings: poses, weights, scorer calls, resampling decisions and the maps of the first and last
    particle bit for bit."""
    a, b = _pf_maps_twin(pkg, True), _pf_maps_twin(pkg, False)
    assert set(a) == set(b) and len(a) == 2 * (4 * 3 + 4)
    for k in a:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
