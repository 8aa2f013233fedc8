"""CPU suite: the C restatement (oracle/slam_oracle.c) against the golden vectors captured from the
compiled reference (tests/golden/, tests/golden/make_golden.py).  Bit-exact unless noted."""
import numpy as np
import pytest
from helpers import SCENES, assert_trace_equal, filtered_scan, load, map_from, trace, trig_scan
from pyoracle import (OIE_DISCREPANCY, OOPE_GMAPPING, OOPE_MAX, OOPE_MEAN, OOPE_OVERLAP, SM_HC,
                      SUM_TREE256, Oracle, ScanData, make_cfg)


def test_enumerator_known_answers(oracle):
    g = load("enumerators.npz")
    for k in ("mc_666666", "mc_42", "hc_2", "hc_6"):
        got = oracle.enumerate_all_rejected(int(g[k + "_kind"]), g[k + "_params"], g[k + "_base"])
        np.testing.assert_array_equal(got, g[k + "_poses"])
    # SURVEY Appendix B pins (captured from the compiled reference)
    p = g["mc_666666_poses"]
    assert p[0].tolist() == [0.19982937958999936, 0.052771745438390721, -0.12569700517621943]
    assert len(g["hc_2_poses"]) == 13 and g["hc_2_poses"][12].tolist() == [0.025, 0.0, 0.0]


def test_oope_known_answers(oracle):
    """The 18 reference OOPE tests (occupancy_observation_probability_test.cpp:59-207)."""
    g = load("oope_known.npz")
    m = map_from(g)
    eps = np.finfo(np.float64).eps
    for kind, lit, ref_out, obst, r in zip(g["kinds"], g["expected_literal"], g["reference_out"],
                                          g["obstacle"], g["range4"]):
        hv, hh = (r[1] - r[0]) / 2, (r[3] - r[2]) / 2
        area = [obst[1] - hv, obst[1] + hv, obst[0] - hh, obst[0] + hh]
        got = oracle.oope_probability(m, make_cfg(oope=int(kind)), obst[0], obst[1], area)
        assert got == ref_out
        assert abs(got - lit) <= eps


@pytest.mark.parametrize("scene", SCENES)
def test_scene_filter_weights_scores(oracle, scene):
    g = load("scene_%s.npz" % scene)
    m = map_from(g)
    tr = trig_scan(g, g["raw_range"], g["raw_angle"])
    kept = oracle.filter_scan(m, g["raw_range"], g["raw_angle"], g["raw_occ"], g["init_pose"],
                              trig=tr)
    np.testing.assert_array_equal(g["raw_range"][kept], g["f_range"])
    np.testing.assert_array_equal(g["raw_angle"][kept], g["f_angle"])
    wname = {0: "even", 1: "viny", 2: "ahr"}[int(g["weighting"])]
    np.testing.assert_array_equal(oracle.weights(wname, g["f_range"], g["f_angle"]), g["f_weight"])
    scan = filtered_scan(g)
    s = oracle.score_poses(m, scan, make_cfg(), g["poses"])
    np.testing.assert_array_equal(s, g["scores"])
    assert s[0] == s[1]
    # canonical tree order of the HIP kernels: same value up to summation rounding
    st = oracle.score_poses(m, scan, make_cfg(sum_order=SUM_TREE256), g["poses"])
    np.testing.assert_allclose(st, g["scores"], rtol=1e-13, atol=0)
    assert st[0] == st[1]
    area = g["win_area"]
    for name, kind in (("max", OOPE_MAX), ("mean", OOPE_MEAN), ("overlap", OOPE_OVERLAP)):
        cfg = make_cfg(oope=kind, area=area)
        sw = oracle.score_poses(m, scan, cfg, g["poses"][:24])
        np.testing.assert_array_equal(sw, g["win_%s_scores" % name])


@pytest.mark.parametrize("scene", SCENES)
@pytest.mark.parametrize("matcher", ["mc", "mc_long", "hc6", "hc128"])
def test_scene_matcher_traces(oracle, scene, matcher):
    g = load("scene_%s.npz" % scene)
    m, scan = map_from(g), filtered_scan(g)
    e = oracle.enumerator(int(g[matcher + "_kind"]), g[matcher + "_params"])
    t = oracle.process_scan(e, m, scan, make_cfg(), g["init_pose"])
    assert_trace_equal(t, trace(g, matcher + "_"))
    if matcher == "mc":  # engine not reseeded between process_scan calls (Q7)
        t2 = oracle.process_scan(e, m, scan, make_cfg(), g["init_pose"])
        assert_trace_equal(t2, trace(g, "mc_second_"))


def test_hc_smoke_cases(oracle):
    """hill_climbing_sm_smoke_test.cpp:72-105: trace parity + the test's own acceptance rule."""
    g = load("hc_smoke.npz")
    m = map_from(g)
    rpose = g["rpose"]
    for i, nz in enumerate(g["noises"]):
        kept = oracle.filter_scan(m, g["raw_range"], g["raw_angle"], g["raw_occ"], rpose + nz)
        scan = ScanData(g["raw_range"][kept], g["raw_angle"][kept])
        e = oracle.enumerator(SM_HC, g["params"])
        t = oracle.process_scan(e, m, scan, make_cfg(), rpose + nz)
        assert_trace_equal(t, trace(g, "case%d_" % i))
        result_noise = nz + t["delta"]
        k0 = oracle.filter_scan(m, g["raw_range"], g["raw_angle"], g["raw_occ"], rpose)
        s0 = ScanData(g["raw_range"][k0], g["raw_angle"][k0])
        p_true = oracle.score_poses(m, s0, make_cfg(), rpose)[0]
        p_res = oracle.score_poses(m, s0, make_cfg(), rpose + result_noise)[0]
        assert p_true == float(g["case%d_prob_true" % i][0])
        assert p_res == float(g["case%d_prob_result" % i][0])
        same_prob = abs(p_true - p_res) <= 1e-7 * max(1.0, abs(p_true), abs(p_res))
        assert same_prob or np.all(np.abs(result_noise) <= np.finfo(np.float64).eps)
        # same case through the cached trig provider
        from pyoracle import TRIG_CACHED
        tr = ScanData(g["raw_range"], g["raw_angle"], None, None, TRIG_CACHED, float(g["a_min"]),
                      float(g["a_inc"]), g["tab_sin"], g["tab_cos"])
        kc = oracle.filter_scan(m, g["raw_range"], g["raw_angle"], g["raw_occ"], rpose + nz, trig=tr)
        sc = ScanData(g["raw_range"][kc], g["raw_angle"][kc], None, None, TRIG_CACHED,
                      float(g["a_min"]), float(g["a_inc"]), g["tab_sin"], g["tab_cos"])
        e = oracle.enumerator(SM_HC, g["params"])
        assert_trace_equal(oracle.process_scan(e, m, sc, make_cfg(), rpose + nz), trace(g, "cached%d_" % i))


def test_gmapping_scene(oracle):
    g = load("gmapping_scene.npz")
    m, scan = map_from(g), filtered_scan(g)
    cfg = make_cfg(oope=OOPE_GMAPPING, oie=OIE_DISCREPANCY)
    cache = Oracle.new_gm_cache()
    s = oracle.score_poses(m, scan, cfg, g["poses"], cache)  # one cache across all poses (Q19)
    # (r06: bit for bit -- until then 1e-13: the restatement evaluated the raw provider's sin / cos pair as ONE sincos(),
    # which in glibc is another build of the functions than the reference's two separate calls; oracle/slam_oracle.c)
    np.testing.assert_array_equal(s, g["scores"])
    kept = oracle.filter_scan(m, g["raw_range"], g["raw_angle"], g["raw_occ"], g["init_pose"],
                              skip_rate=3)
    np.testing.assert_array_equal(g["raw_range"][kept], g["skip3_range"])
    s3 = ScanData(g["skip3_range"], g["skip3_angle"])
    e = oracle.enumerator(SM_HC, [6, 0.1, 0.1])
    t = oracle.process_scan(e, m, s3, cfg, g["init_pose"], cache=Oracle.new_gm_cache())
    assert_trace_equal(t, trace(g, "hc6_skip3_"))


def test_resample_golden(oracle):
    g = load("resample.npz")
    for k in range(int(g["n_cases"])):
        w, seed = g["w%d" % k], int(g["seed%d" % k])
        np.testing.assert_array_equal(oracle.resample(w, seed), g["idx%d" % k])
        assert oracle.resampling_is_required(w) == bool(int(g["req%d" % k]))


def test_weights_and_filter_golden(oracle):
    g = load("weights_ahr.npz")
    for name in ("even", "viny", "ahr"):
        np.testing.assert_array_equal(oracle.weights(name, g["range"], g["angle"]), g["w_" + name])
    from pyoracle import CELL_OCC, GridMapData
    m = GridMapData(CELL_OCC, np.full((400, 400), 0.5), (200, 200), 0.1, [0.5])
    kept = oracle.filter_scan(m, g["range"], g["angle"], g["occ"], g["filt_pose"], skip_rate=3,
                              max_range=4.5)
    np.testing.assert_array_equal(g["range"][kept], g["filt_range"])
    mb = GridMapData(CELL_OCC, np.full((60, 60), 0.5), (30, 30), 0.1, [0.5], bounded=True)
    kept = oracle.filter_scan(mb, g["range"], g["angle"], g["occ"], g["filt_pose"])
    np.testing.assert_array_equal(g["range"][kept], g["filt_bounded_range"])


def test_world_to_cells_golden(oracle):
    g = load("world_to_cells.npz")
    for i, s in enumerate(g["segments"]):
        got = oracle.world_to_cells(float(g["scale"]), *s)
        np.testing.assert_array_equal(got, g["cells"][g["offsets"][i]:g["offsets"][i + 1]])


@pytest.mark.parametrize("scenario", ["default", "nogate", "wide"])
def test_gmapping_particle_filter_steps(oracle, scenario):
    """G5: GmappingParticleFilter::handle_sensor_data over several scans (gate, pose noise, HC
    match with the shared OOPE cache, weights, N_eff test, resampling with duplicated particles and
    master hand-over) against the compiled reference; map update off via max_range = 0."""
    g = load("gmapping_pf.npz")
    m = map_from(g, scenario + "_map_")
    n = len(g[scenario + "_seeds"])
    pf = oracle.gmapping_create(n, g[scenario + "_gp"], g[scenario + "_seeds"], skip_rate=3)
    any_resampled = False
    for k in range(int(g[scenario + "_n_steps"])):
        pre = "%s_step%d_" % (scenario, k)
        extra = np.arange(5000 + 100 * k, 5000 + 100 * k + n, dtype=np.uint32)
        res, _idx = pf.step(m, g[pre + "range"], g[pre + "angle"], None, g[pre + "delta"], 7 + k, extra)
        poses, w, ms = pf.state()
        assert res == bool(int(g[pre + "resampled"])), k
        np.testing.assert_array_equal(ms, g[pre + "master"])
        np.testing.assert_array_equal(poses, g[pre + "poses"])
        np.testing.assert_array_equal(w, g[pre + "weights"])
        any_resampled |= res
    if scenario == "wide":
        assert any_resampled
