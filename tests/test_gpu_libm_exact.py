"""GPU suite: the libm-exact modes (csrc/libm_exact.h, csrc/exact_kernels.hip).

The reference's DEFAULT trig provider is RawTrigonometryProvider -- std::sin / std::cos(theta + a) per beam and pose
(src/core/trigonometry_utils.h:17-35; use_trig_cache = false, src/ros/init_utils.h:56-58) -- and its GMapping cell's
discrepancy is 1 - std::exp(-d^2 / 0.05) (src/slams/gmapping/gmapping_grid_cell.h:35-38).  Every other mode of the
library evaluates the cached provider's angle addition and the device's own exp: inside the 1e-5 contract, not the
reference's bits.  Here the device evaluates glibc's functions restated operation for operation, and the bar is
assert_array_equal against the compiled reference's own outputs (tests/golden/*.npz, RAW provider):
  * the device's sin / cos / exp == the host libm's, bit for bit (the CPU suite holds the host restatement to libm over
    > 10^8 arguments: tests/test_libm_exact.py);
  * POSE_TRIG_RAW_EXACT: the raw-provider scenes, and the reference's HC smoke fixture -- whose end points sit EXACTLY on
    cell edges, so that the last place of sin(theta + a) picks the cell -- trace for trace;
  * the GMapping OOPE in its exact mode: scores of a call sequence with the cache carried across poses, the cache left
    behind, an HC(6) trace, and whole filter runs (poses and weights of every particle after every step)."""
import ctypes as C
import ctypes.util

import numpy as np
import pytest
from helpers import assert_trace_equal, filtered_scan, load, map_from, trace

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
def variant(pkg):
    v = pkg.libm_variant()
    if v < 0:
        pytest.skip("this host's libm is neither build of glibc's sin / cos / exp: no exact modes here")
    return v


def test_device_functions_return_the_host_libms_bits(pkg, ctx, variant):
    libm = C.CDLL(ctypes.util.find_library("m") or "libm.so.6")
    rs = np.random.RandomState(12)
    n = 60000
    trig = np.concatenate([rs.uniform(-7, 7, n), rs.uniform(-0.9, 0.9, n), rs.uniform(-3000, 3000, n), rs.uniform(-1e8, 1e8, n),
                           np.arange(-900, 900) / 128.0, np.arange(-4000, 4000) * (np.pi / 2), [0.0, -0.0, 1e-300, 0.126, 0.855469, 2.426265]])
    ex = np.concatenate([-rs.uniform(0, 2, n), -rs.uniform(0, 40, n), rs.uniform(-745, 709, n), rs.uniform(-1100, -500, 2000),
                         [0.0, -0.0, 1e-320, -1e-320, -745.2, -1024.0, 709.7, 1024.0, 512.0, -512.0, np.inf, -np.inf]])
    for fn, name, x in ((0, "sin", trig), (1, "cos", trig), (2, "exp", ex)):
        f = getattr(libm, name)
        f.restype, f.argtypes = C.c_double, [C.c_double]
        want = np.array([f(float(v)) for v in x])
        got = ctx.libm_eval(variant, fn, x)
        bad = np.nonzero(got.view(np.uint64) != want.view(np.uint64))[0]
        assert len(bad) == 0, (name, x[bad[:5]], got[bad[:5]], want[bad[:5]])


def upload_raw(pkg, ctx, m, scan, map_id=0):
    ctx.upload_map(map_id, m)
    cos_a, sin_a = pkg.beam_trig(scan.angle)
    ctx.scan_upload(scan.range, cos_a, sin_a, scan.weight, scan.factor)
    ctx.scan_set_angles(scan.angle)


@pytest.mark.parametrize("scene", ["mean_raw", "tbm_raw", "affine_raw"])
def test_raw_provider_scenes_bit_for_bit(pkg, ctx, variant, scene):
    g = load("scene_%s.npz" % scene)
    m, scan = map_from(g), filtered_scan(g)
    upload_raw(pkg, ctx, m, scan)
    strict = pkg.spe_cfg(sum_order=pkg.SUM_SEQUENTIAL, pose_trig=pkg.POSE_TRIG_RAW_EXACT)
    np.testing.assert_array_equal(ctx.score_poses(0, strict, g["poses"]), g["scores"])
    # the canonical sum over the raw provider's end points: 1e-12, and bit-equal for bit-equal poses
    tree = ctx.score_poses(0, pkg.spe_cfg(pose_trig=pkg.POSE_TRIG_RAW_EXACT), g["poses"])
    np.testing.assert_allclose(tree, g["scores"], rtol=1e-12, atol=0)
    assert tree[0] == tree[1]
    for name in ("mc", "hc6", "hc128"):
        mt = pkg.Matcher(ctx, "MC" if name == "mc" else "HC", strict, g[name + "_params"])
        assert_trace_equal(mt.process_scan(0, g["init_pose"], trace=True), trace(g, name + "_"))
    if "win_area" in g:
        for kname, kk in (("max", pkg.OOPE_MAX), ("mean", pkg.OOPE_MEAN), ("overlap", pkg.OOPE_OVERLAP)):
            cfg = pkg.spe_cfg(oope=kk, area=tuple(g["win_area"]), sum_order=pkg.SUM_SEQUENTIAL, pose_trig=pkg.POSE_TRIG_RAW_EXACT)
            np.testing.assert_array_equal(ctx.score_poses(0, cfg, g["poses"][:24]), g["win_" + kname + "_scores"])


def test_hc_smoke_fixture_with_the_raw_provider_trace_for_trace(pkg, ctx, variant):
    """hill_climbing_sm_smoke_test.cpp:72-105: robot on a cell centre, steps of half a cell -- end points exactly on cell
    edges, where the last place of libm's sin(theta + a) decides the cell.  r01-r05 met the reference test's own
    acceptance rule there; with the restated libm the RAW-provider trace is the reference's, bit for bit."""
    g = load("hc_smoke.npz")
    m = map_from(g)
    ctx.upload_map(0, m)
    geom = dict(width=m.width, height=m.height, origin=m.origin, scale=m.scale, bounded=False)
    strict = pkg.spe_cfg(sum_order=pkg.SUM_SEQUENTIAL, pose_trig=pkg.POSE_TRIG_RAW_EXACT)
    checked = 0
    for i, nz in enumerate(g["noises"]):
        pose = g["rpose"] + nz
        kept = pkg.filter_scan(g["raw_range"], g["raw_angle"], g["raw_occ"], pose, geom)
        r, a = g["raw_range"][kept], g["raw_angle"][kept]
        c, s = pkg.beam_trig(a)
        ctx.scan_upload(r, c, s, pkg.scan_weights("even", r, a))
        ctx.scan_set_angles(a)
        mt = pkg.Matcher(ctx, "HC", strict, g["params"])
        assert_trace_equal(mt.process_scan(0, pose, trace=True), trace(g, "case%d_" % i))
        checked += 1
    assert checked == len(g["noises"])


def test_gmapping_oope_scores_cache_and_trace_bit_for_bit(pkg, ctx, variant):
    g = load("gmapping_scene.npz")
    m, scan = map_from(g), filtered_scan(g)
    upload_raw(pkg, ctx, m, scan)
    cfg = pkg.spe_cfg(oope=pkg.OOPE_GMAPPING, sum_order=pkg.SUM_SEQUENTIAL, pose_trig=pkg.POSE_TRIG_RAW_EXACT)
    ctx.gm_cache_reset()
    s = ctx.score_poses(0, cfg, g["poses"])  # ONE call sequence: the cache carries across poses (Q19)
    np.testing.assert_array_equal(s, g["scores"])
    cache = ctx.gm_cache_get()
    for split in (17, 1, 40):  # the carry survives the call boundary
        ctx.gm_cache_reset()
        s2 = np.concatenate([ctx.score_poses(0, cfg, g["poses"][:split]), ctx.score_poses(0, cfg, g["poses"][split:])])
        np.testing.assert_array_equal(s2, g["scores"])
        assert ctx.gm_cache_get() == cache
    r3, a3 = g["skip3_range"], g["skip3_angle"]
    c, sn = pkg.beam_trig(a3)
    ctx.scan_upload(r3, c, sn, pkg.scan_weights("even", r3, a3))
    ctx.scan_set_angles(a3)
    ctx.gm_cache_reset()
    mt = pkg.Matcher(ctx, "HC", cfg, [6, 0.1, 0.1])
    assert_trace_equal(mt.process_scan(0, g["init_pose"], trace=True), trace(g, "hc6_skip3_"))


@pytest.mark.parametrize("scenario", ["default", "nogate", "wide"])
def test_gmapping_filter_runs_bit_for_bit(pkg, ctx, variant, scenario):
    """tests/golden/gmapping_pf.npz (the compiled reference's GmappingParticleFilter over several scans, raw provider):
    every particle's pose and weight after every step, assert_array_equal."""
    g = load("gmapping_pf.npz")
    m = map_from(g, scenario + "_map_")
    ctx.upload_map(5, m)
    n = len(g[scenario + "_seeds"])
    pf = pkg.GmappingFilter(ctx, pkg.gmapping_params(gp8=g[scenario + "_gp"], skip_rate=3, pose_trig=pkg.POSE_TRIG_RAW_EXACT),
                            n, g[scenario + "_seeds"])
    for k in range(int(g[scenario + "_n_steps"])):
        pre = "%s_step%d_" % (scenario, k)
        res, _idx = pf.step(5, g[pre + "range"], g[pre + "angle"], None, g[pre + "delta"], 7 + k)
        poses, w, ms = pf.state()
        assert res == bool(int(g[pre + "resampled"])), k
        np.testing.assert_array_equal(ms, g[pre + "master"])
        np.testing.assert_array_equal(poses, g[pre + "poses"])
        np.testing.assert_array_equal(w, g[pre + "weights"])
    ctx.map_release(5)


def test_exact_modes_say_what_they_need(pkg, ctx, variant):
    g = load("scene_mean_raw.npz")
    m, scan = map_from(g), filtered_scan(g)
    ctx.upload_map(0, m)
    cos_a, sin_a = pkg.beam_trig(scan.angle)
    ctx.scan_upload(scan.range, cos_a, sin_a, scan.weight, scan.factor)  # no angles
    with pytest.raises(pkg.SlamHipError, match="scan_set_angles"):
        ctx.score_poses(0, pkg.spe_cfg(pose_trig=pkg.POSE_TRIG_RAW_EXACT), g["poses"][:2])
    with pytest.raises(pkg.SlamHipError):
        ctx.scan_set_angles(scan.angle[:-1])


# ---- the map update with the raw provider (slamhip_map_append_scan_raw) ------------------------------------------------
MODELS = {"mean": (0, 2), "affine": (0, 1), "last": (0, 0), "tbm": (1, 3), "gmapping": (2, 4)}  # (cell model, rule)
STRIDE = {0: 1, 1: 4, 2: 3}
AUX = {2: 1, 4: 2}
K6_PATHS = {"gather": 0, "counting": 1, "radix": 2}


@pytest.mark.parametrize("path", list(K6_PATHS))
@pytest.mark.parametrize("name", list(MODELS))
@pytest.mark.parametrize("golden,estimator", [("map_update.npz", 0), ("map_update_area.npz", 1)])
def test_append_scan_with_the_raw_provider_bit_for_bit(pkg, ctx, variant, golden, estimator, name, path):
    """tests/golden/map_update.npz and map_update_area.npz hold the compiled reference's maps after every append_scan
    with its default RawTrigonometryProvider.  tests/test_gpu_mapupdate.py meets them to 1e-11 .. 1e-13 with the cached
    provider's angle addition (obstacle means and the area estimator's split depend continuously on the end point);
    with the raw provider's own cos / sin(theta + a): every payload double and every counter, assert_array_equal."""
    g = load(golden)
    if name + "_size" not in g:
        pytest.skip("%s holds no %s map" % (golden, name))
    ctx.set_option(pkg.OPT_K6_PATH, K6_PATHS[path])
    try:
        cell_model, rule = MODELS[name]
        w, h = [int(v) for v in g[name + "_size"]]
        st = STRIDE[cell_model]
        ctx.map_bind(2, cell_model, w, h, g[name + "_origin"], float(g["scale"]), g[name + "_unknown"][:st])
        lo, hi = [int(v) for v in g["crop"]]
        for k in range(int(g["n_steps"])):
            q, blur, max_range = g["step%d_params" % k]
            extra = dict(estimator=1, shift_amount=float(g["shift_amount"])) if estimator else {}
            ctx.map_append_scan_raw(2, rule, g["step%d_pose" % k], g["step%d_range" % k], g["step%d_angle" % k],
                                    g["step%d_occ" % k], quality=q, base=g[name + "_base"], blur=blur, max_range=max_range,
                                    **extra)
            got = ctx.map_download_window(2, lo, lo, hi - lo, hi - lo, st)
            np.testing.assert_array_equal(got, g["%s_step%d_payload" % (name, k)], err_msg="%s step %d" % (name, k))
            if rule in AUX:
                np.testing.assert_array_equal(ctx.map_download_aux(2, lo, lo, hi - lo, hi - lo, AUX[rule]),
                                              g["%s_step%d_aux" % (name, k)])
        ctx.map_release(2)
    finally:
        ctx.set_option(pkg.OPT_K6_PATH, 0)


def test_full_gmapping_step_with_map_update_bit_for_bit(pkg, ctx, variant):
    """tests/golden/gmapping_pf_update.npz: the reference's FULL GMapping step -- every matching particle appends its scan
    to the one shared map before the next particle matches -- over several scans: poses, weights, the whole map and
    its counters after every step, assert_array_equal."""
    g = load("gmapping_pf_update.npz")
    w, h = [int(v) for v in g["size"]]
    ctx.map_bind(4, 2, w, h, g["origin"], float(g["scale"]), g["unknown"][:3])
    n = len(g["seeds"])
    pf = pkg.GmappingFilter(ctx, pkg.gmapping_params(gp8=g["gp"], skip_rate=3, pose_trig=pkg.POSE_TRIG_RAW_EXACT), n, g["seeds"])
    pf.set_map_update(True)
    for k in range(int(g["n_steps"])):
        res, _ = pf.step(4, g["step%d_range" % k], g["step%d_angle" % k], None, g["step%d_delta" % k], 7 + k)
        poses, wts, ms = pf.state()
        assert res == bool(int(g["step%d_resampled" % k]))
        np.testing.assert_array_equal(ms, g["step%d_master" % k])
        np.testing.assert_array_equal(poses, g["step%d_poses" % k])
        np.testing.assert_array_equal(wts, g["step%d_weights" % k])
        np.testing.assert_array_equal(ctx.map_download_window(4, 0, 0, w, h, 3), g["step%d_payload" % k])
        np.testing.assert_array_equal(ctx.map_download_aux(4, 0, 0, w, h, 2), g["step%d_aux" % k])
    ctx.map_release(4)
