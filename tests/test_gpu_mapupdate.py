"""GPU suite: K6 map update (slamhip_map_append_scan) through the C-ABI against the payloads and
update counters exported from the compiled reference after every append_scan
(tests/golden/map_update.npz), and against the oracle at BASELINE size.  Bit-exact."""
import numpy as np
import pytest
from helpers import load

import __graft_entry__ as ge

pytestmark = pytest.mark.gpu

MODELS = {"mean": (0, 2), "affine": (0, 1), "last": (0, 0), "tbm": (1, 3), "gmapping": (2, 4)}  # (cell model, rule)
STRIDE = {0: 1, 1: 4, 2: 3}
AUX = {2: 1, 4: 2}


@pytest.fixture(scope="module")
def pkg():
    return ge.load_package()


@pytest.fixture(scope="module")
def ctx(pkg):
    c = pkg.Context(0)
    yield c
    c.close()


K6_PATHS = {"gather": 0, "counting": 1, "radix": 2}


@pytest.fixture(autouse=True)
def _default_k6_path(pkg, ctx):
    set_k6_path(pkg, ctx, "gather")
    yield


def set_k6_path(pkg, ctx, name):
    ctx.set_option(pkg.OPT_K6_PATH, K6_PATHS[name])


@pytest.mark.parametrize("path", list(K6_PATHS))
@pytest.mark.parametrize("name", list(MODELS))
def test_append_scan_vs_reference_golden(pkg, ctx, name, path):
    set_k6_path(pkg, ctx, path)
    g = load("map_update.npz")
    cell_model, rule = MODELS[name]
    w, h = [int(v) for v in g[name + "_size"]]
    st = STRIDE[cell_model]
    unk = g[name + "_unknown"]
    ctx.map_bind(2, cell_model, w, h, g[name + "_origin"], float(g["scale"]), unk[:st])
    lo, hi = [int(v) for v in g["crop"]]
    for k in range(int(g["n_steps"])):
        q, blur, max_range = g["step%d_params" % k]
        c, s = pkg.beam_trig(g["step%d_angle" % k])
        nu = ctx.map_append_scan(2, rule, g["step%d_pose" % k], g["step%d_range" % k], c, s, g["step%d_occ" % k],
                                 quality=q, base=g[name + "_base"], blur=blur, max_range=max_range)
        assert nu > 1000
        got = ctx.map_download_window(2, lo, lo, hi - lo, hi - lo, st)
        want = g["%s_step%d_payload" % (name, k)]
        if name == "gmapping":
            # obstacle means accumulate endpoints: the raw trig provider's libm sin(theta + a) and the
            # device's angle-addition form differ in the last ulp
            np.testing.assert_array_equal(got[..., 0], want[..., 0])
            np.testing.assert_allclose(got[..., 1:], want[..., 1:], rtol=1e-13, atol=1e-15)
        else:
            np.testing.assert_array_equal(got, want, err_msg="%s step %d" % (name, k))
        if rule in AUX:
            np.testing.assert_array_equal(ctx.map_download_aux(2, lo, lo, hi - lo, hi - lo, AUX[rule]),
                                          g["%s_step%d_aux" % (name, k)])
    ctx.map_release(2)


def test_append_scan_full_size_vs_oracle_and_rescoring(pkg, ctx):
    """1080 beams on a 2000x2000 @0.05 m map: update on the GPU == update by the oracle, the updated
    map scores like the oracle's, and a beam leaving the window is reported."""
    import pyoracle as po
    from pyoracle_mapupdate import RULE_MEAN, append_scan
    from synth import make_scene
    O = po.Oracle()
    sc = make_scene(cell_model=0, size=2000, scale=0.05, n_beams=1080, seed=21)
    m, scan = sc["map"], sc["scan"]
    ctx.upload_map(2, m)
    aux = np.zeros((m.height, m.width, 1))
    rs = np.random.RandomState(3)
    for k in range(3):
        pose = sc["true_pose"] + rs.randn(3) * [0.2, 0.2, 0.05]
        c, s = pkg.beam_trig(scan.angle)
        # cached-provider arithmetic on both sides: tabulate the provider with one entry per beam
        tr = po.ScanData(scan.range, scan.angle, None, None, po.TRIG_CACHED, 0.0, 1.0, s, c)
        tr.angle = np.arange(scan.n, dtype=np.float64)  # index == beam
        nu_o = append_scan(O, m, aux, RULE_MEAN, pose, scan.range, tr.angle, None, quality=0.9, blur=0.3, trig=tr)
        nu = ctx.map_append_scan(2, pkg.RULE_MEAN, pose, scan.range, c, s, None, quality=0.9, blur=0.3)
        assert nu == nu_o
    np.testing.assert_array_equal(ctx.map_download_window(2, 0, 0, m.width, m.height, 1), m.payload)
    np.testing.assert_array_equal(ctx.map_download_aux(2, 0, 0, m.width, m.height, 1), aux)
    ctx.scan_upload(scan.range, c, s, scan.weight, scan.factor)
    poses = sc["init_pose"] + rs.randn(64, 3) * [0.1, 0.1, 0.05]
    got = ctx.score_poses(2, pkg.spe_cfg(sum_order=1, pose_trig=1), poses)
    np.testing.assert_array_equal(got, O.score_poses(m, scan, po.make_cfg(), poses))
    with pytest.raises(pkg.SlamHipError):
        ctx.map_append_scan(2, pkg.RULE_MEAN, [49.0, 0.0, 0.0], scan.range, c, s)
    ctx.map_release(2)


@pytest.mark.parametrize("path", list(K6_PATHS))
@pytest.mark.parametrize("name", ["mean", "tbm", "gmapping"])
def test_append_scan_area_estimator_vs_reference_golden(pkg, ctx, name, path):
    set_k6_path(pkg, ctx, path)
    _area_estimator_vs_reference_golden(pkg, ctx, name)


def _area_estimator_vs_reference_golden(pkg, ctx, name):
    """AreaOccupancyEstimator on the GPU (slam/occupancy_estimator/type = area, BASELINE cfg 5)."""
    g = load("map_update_area.npz")
    cell_model, rule = MODELS[name]
    w, h = [int(v) for v in g[name + "_size"]]
    st = STRIDE[cell_model]
    ctx.map_bind(2, cell_model, w, h, g[name + "_origin"], float(g["scale"]), g[name + "_unknown"][:st])
    lo, hi = [int(v) for v in g["crop"]]
    for k in range(int(g["n_steps"])):
        q, blur, max_range = g["step%d_params" % k]
        c, s = pkg.beam_trig(g["step%d_angle" % k])
        ctx.map_append_scan(2, rule, g["step%d_pose" % k], g["step%d_range" % k], c, s, g["step%d_occ" % k],
                            quality=q, base=g[name + "_base"], blur=blur, max_range=max_range, estimator=1,
                            shift_amount=float(g["shift_amount"]))
        got = ctx.map_download_window(2, lo, lo, hi - lo, hi - lo, st)
        want = g["%s_step%d_payload" % (name, k)]
        # the area split depends continuously on the endpoint: raw-provider endpoints differ from the
        # reference's libm sin(theta + a) in the last ulp -> 1e-12, not bits
        np.testing.assert_allclose(got, want, rtol=1e-11, atol=1e-13, err_msg="%s step %d" % (name, k))
        if rule in AUX:
            np.testing.assert_array_equal(ctx.map_download_aux(2, lo, lo, hi - lo, hi - lo, AUX[rule]),
                                          g["%s_step%d_aux" % (name, k)])
    ctx.map_release(2)


@pytest.mark.parametrize("pose_trig", [1, 0])
def test_gmapping_filter_with_map_update_vs_reference_golden(pkg, ctx, pose_trig):
    """The reference's FULL GMapping step through the C-ABI: every matching particle appends its scan
    to the one shared map before the next particle matches (slamhip_gmapping_set_map_update).
    pose_trig 1 = host trigonometry (host-driven matches, bit-exact mode); 0 = the default: every match is a chain
    of kernels on the device (GMapping OOPE, hc_chain.hip) and the map updates are queued behind them without
    waiting -- same particles, same map."""
    g = load("gmapping_pf_update.npz")
    w, h = [int(v) for v in g["size"]]
    ctx.map_bind(4, 2, w, h, g["origin"], float(g["scale"]), g["unknown"][:3])
    n = len(g["seeds"])
    pf = pkg.GmappingFilter(ctx, pkg.gmapping_params(gp8=g["gp"], skip_rate=3, pose_trig=pose_trig), n, g["seeds"])
    pf.set_map_update(True)
    for k in range(int(g["n_steps"])):
        res, _ = pf.step(4, g["step%d_range" % k], g["step%d_angle" % k], None, g["step%d_delta" % k], 7 + k)
        poses, wts, ms = pf.state()
        assert res == bool(int(g["step%d_resampled" % k]))
        np.testing.assert_array_equal(ms, g["step%d_master" % k])
        np.testing.assert_allclose(poses, g["step%d_poses" % k], rtol=0, atol=1e-10)
        np.testing.assert_allclose(wts, g["step%d_weights" % k], rtol=1e-9, atol=0)
        got = ctx.map_download_window(4, 0, 0, w, h, 3)
        want = g["step%d_payload" % k]
        np.testing.assert_array_equal(got[..., 0], want[..., 0])  # occupancy: bit-exact
        np.testing.assert_allclose(got[..., 1:], want[..., 1:], rtol=1e-12, atol=1e-14)  # obstacle means
        np.testing.assert_array_equal(ctx.map_download_aux(4, 0, 0, w, h, 2), g["step%d_aux" % k])
    ctx.map_release(4)


def test_counting_sorted_and_radix_sorted_updates_agree(pkg, ctx):
    """A plain update is counting-sorted over the window around the scan; scans of more than 4096 beams and windows
    of more than 2^23 cells take the radix-sorted path (the fall-back every update took in round 1).  The same
    beams, appended once as ONE scan of 5400 beams (radix) and once as five scans of 1080 (counting) in beam
    order, must leave the same map for the order-insensitive part of the state -- GMapping hit / try counters
    are sums -- and a long-range scan whose window exceeds 2^23 cells must equal the oracle."""
    import pyoracle as po
    from pyoracle_mapupdate import RULE_MEAN, append_scan
    from synth import make_scene
    sc = make_scene(cell_model=2, size=1200, scale=0.05, n_beams=1080, seed=31)
    m, scan = sc["map"], sc["scan"]
    pose = sc["true_pose"]
    unknown = [-1.0, 0.0, 0.0]
    rs = np.random.RandomState(5)
    ranges = [np.clip(scan.range * (0.6 + 0.1 * k) + rs.rand(scan.n) * 0.2, 0.3, 25.0) for k in range(5)]
    c, s = pkg.beam_trig(scan.angle)
    for mid in (2, 3):
        ctx.map_bind(mid, 2, m.width, m.height, m.origin, m.scale, unknown)
    nu_small = sum(ctx.map_append_scan(2, pkg.RULE_GMAPPING, pose, r, c, s, None) for r in ranges)
    nu_big = ctx.map_append_scan(3, pkg.RULE_GMAPPING, pose, np.concatenate(ranges), np.tile(c, 5), np.tile(s, 5), None)
    assert nu_small == nu_big > 5 * 1080 * 20
    np.testing.assert_array_equal(ctx.map_download_aux(2, 0, 0, m.width, m.height, 2),
                                  ctx.map_download_aux(3, 0, 0, m.width, m.height, 2))  # hits, tries
    a = ctx.map_download_window(2, 0, 0, m.width, m.height, 3)
    b = ctx.map_download_window(3, 0, 0, m.width, m.height, 3)
    np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-15)  # running means: same observations, same order per cell
    ctx.map_release(2)
    ctx.map_release(3)
    # a window of more than 2^23 cells: 3100 x 3100 cells between the robot and the farthest end points
    O = po.Oracle()
    size = 3200
    payload = np.full((size, size, 1), 0.5)
    big = po.GridMapData(0, payload, (size // 2, size // 2), 0.05, [0.5])
    aux = np.zeros((size, size, 1))
    ctx.upload_map(2, big)
    ang = np.deg2rad(np.linspace(-180, 180, 720, endpoint=False))
    rng = np.full(ang.size, 77.0)  # 1540 cells along the axes: a window of 3081 x 3081 cells
    rng[::7] = 55.5
    c, s = pkg.beam_trig(ang)
    tr = po.ScanData(rng, ang, None, None, po.TRIG_CACHED, 0.0, 1.0, s, c)
    tr.angle = np.arange(ang.size, dtype=np.float64)
    p0 = np.array([0.31, -0.17, 0.2])
    nu_o = append_scan(O, big, aux, RULE_MEAN, p0, rng, tr.angle, None, quality=0.9, blur=0.0, trig=tr)
    nu = ctx.map_append_scan(2, pkg.RULE_MEAN, p0, rng, c, s, None, quality=0.9, blur=0.0)
    assert nu == nu_o
    np.testing.assert_array_equal(ctx.map_download_window(2, 0, 0, size, size, 1), big.payload)
    ctx.map_release(2)


@pytest.mark.parametrize("sort", list(K6_PATHS))
@pytest.mark.parametrize("name", ["mean", "affine", "tbm", "gmapping"])
def test_append_scan_with_per_point_quality_vs_reference_golden(pkg, name, sort):
    """The `ahr` observation quality estimator (AngleHistogramResiprocalOMQE, grid_map_scan_adders.h:32-43; selected by
    slam/mapping/observation_quality_estimator/typetype, init_occupancy_mapping.h:64-80): K6 with a per-point quality
    (slamhip_map_append_scan_q + slamhip_omqe_quality) against maps exported from the compiled reference running that
    estimator (tests/golden/make_golden_omqe.py), both sorting paths, bit for bit (GMapping obstacle means to the
    raw-provider ulp of the idle golden)."""
    g = load("map_update_ahr.npz")
    cell_model, rule = MODELS[name]
    w, h = [int(v) for v in g[name + "_size"]]
    st = STRIDE[cell_model]
    unk = g[name + "_unknown"]
    ctx = pkg.Context(0)
    set_k6_path(pkg, ctx, sort)
    ctx.map_bind(2, cell_model, w, h, g[name + "_origin"], float(g["scale"]), unk[:st])
    lo, hi = [int(v) for v in g["crop"]]
    for k in range(int(g["n_steps"])):
        q, blur, max_range = g["step%d_params" % k]
        c, s = pkg.beam_trig(g["step%d_angle" % k])
        bq = pkg.omqe_quality(1, g["step%d_range" % k], g["step%d_angle" % k])
        np.testing.assert_array_equal(bq, g["step%d_quality" % k])
        ctx.map_append_scan(2, rule, g["step%d_pose" % k], g["step%d_range" % k], c, s, g["step%d_occ" % k], quality=q,
                            base=g[name + "_base"], blur=blur, max_range=max_range, beam_quality=bq)
        got = ctx.map_download_window(2, lo, lo, hi - lo, hi - lo, st)
        want = g["%s_step%d_payload" % (name, k)]
        if name == "gmapping":
            np.testing.assert_array_equal(got[..., 0], want[..., 0])
            np.testing.assert_allclose(got[..., 1:], want[..., 1:], rtol=1e-13, atol=1e-15)
        else:
            np.testing.assert_array_equal(got, want, err_msg="%s step %d" % (name, k))
        if rule in AUX:
            np.testing.assert_array_equal(ctx.map_download_aux(2, lo, lo, hi - lo, hi - lo, AUX[rule]),
                                          g["%s_step%d_aux" % (name, k)])
    ctx.close()
