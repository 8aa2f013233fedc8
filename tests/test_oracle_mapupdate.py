"""CPU suite: the map-update restatement (oracle/map_update_oracle.c) against payloads and update
counters exported from the compiled reference after every append_scan (tests/golden/map_update.npz,
tests/golden/make_golden_mapupdate.py).  Bit-exact for all five cell update rules."""
import numpy as np
import pytest
from helpers import load
from pyoracle import CELL_GMAPPING, CELL_OCC, CELL_TBM, GridMapData
from pyoracle_mapupdate import (AUX_STRIDE, RULE_AFFINE, RULE_GMAPPING, RULE_LAST, RULE_MEAN, RULE_TBM,
                                append_scan)

MODELS = {"mean": (CELL_OCC, RULE_MEAN), "affine": (CELL_OCC, RULE_AFFINE), "last": (CELL_OCC, RULE_LAST),
          "tbm": (CELL_TBM, RULE_TBM), "gmapping": (CELL_GMAPPING, RULE_GMAPPING)}


def fresh_map(g, name):
    cell_model, rule = MODELS[name]
    w, h = g[name + "_size"]
    st = {CELL_OCC: 1, CELL_TBM: 4, CELL_GMAPPING: 3}[cell_model]
    unk = g[name + "_unknown"]
    payload = np.tile(unk[:st], (h, w, 1)).astype(np.float64)
    m = GridMapData(cell_model, payload, g[name + "_origin"], float(g["scale"]), unk[:st])
    aux = np.zeros((h, w, AUX_STRIDE[rule])) if rule in AUX_STRIDE else None
    return m, aux, rule


@pytest.mark.parametrize("name", list(MODELS))
def test_append_scan_vs_reference(oracle, name):
    g = load("map_update.npz")
    m, aux, rule = fresh_map(g, name)
    lo, hi = g["crop"]
    for k in range(int(g["n_steps"])):
        q, blur, max_range = g["step%d_params" % k]
        n_upd = append_scan(oracle, m, aux, rule, g["step%d_pose" % k], g["step%d_range" % k],
                            g["step%d_angle" % k], g["step%d_occ" % k], quality=q, base=g[name + "_base"],
                            blur=blur, max_range=max_range)
        assert n_upd > 1000
        np.testing.assert_array_equal(m.payload[lo:hi, lo:hi], g["%s_step%d_payload" % (name, k)], err_msg="step %d" % k)
        if aux is not None:
            np.testing.assert_array_equal(aux[lo:hi, lo:hi], g["%s_step%d_aux" % (name, k)])


@pytest.mark.parametrize("name", ["mean", "tbm", "gmapping"])
def test_append_scan_area_estimator_vs_reference(oracle, name):
    """AreaOccupancyEstimator (area_occupancy_estimator.h:27-240): triangle / trapezoid area split,
    segment classification with fuzzy comparisons, the edge-shift static (Q27)."""
    from pyoracle_mapupdate import append_scan_ex
    g = load("map_update_area.npz")
    m, aux, rule = fresh_map(g, name)
    lo, hi = g["crop"]
    for k in range(int(g["n_steps"])):
        q, blur, max_range = g["step%d_params" % k]
        append_scan_ex(oracle, m, aux, rule, g["step%d_pose" % k], g["step%d_range" % k], g["step%d_angle" % k],
                       g["step%d_occ" % k], quality=q, base=g[name + "_base"], blur=blur, max_range=max_range,
                       est_kind=1, shift_amount=float(g["shift_amount"]))
        np.testing.assert_array_equal(m.payload[lo:hi, lo:hi], g["%s_step%d_payload" % (name, k)],
                                      err_msg="step %d" % k)
        if aux is not None:
            np.testing.assert_array_equal(aux[lo:hi, lo:hi], g["%s_step%d_aux" % (name, k)])


def test_gmapping_filter_with_map_update_vs_reference(oracle):
    """The full GMapping step of the reference (every particle appends its scan to the ONE shared
    map before the next particle matches) over five scans: poses, weights, master flags and the
    map's payload + (hits, tries) after every step."""
    from pyoracle_mapupdate import gmapping_enable_update
    g = load("gmapping_pf_update.npz")
    w, h = [int(v) for v in g["size"]]
    payload = np.tile(g["unknown"][:3], (h, w, 1)).astype(np.float64)
    m = GridMapData(CELL_GMAPPING, payload, g["origin"], float(g["scale"]), g["unknown"][:3])
    aux = np.zeros((h, w, 2))
    n = len(g["seeds"])
    pf = oracle.gmapping_create(n, g["gp"], g["seeds"], skip_rate=3)
    gmapping_enable_update(oracle, pf, m, aux)
    for k in range(int(g["n_steps"])):
        extra = np.arange(9000 + 100 * k, 9000 + 100 * k + n, dtype=np.uint32)
        res, _ = pf.step(m, g["step%d_range" % k], g["step%d_angle" % k], None, g["step%d_delta" % k], 7 + k, extra)
        poses, wts, ms = pf.state()
        assert res == bool(int(g["step%d_resampled" % k]))
        np.testing.assert_array_equal(ms, g["step%d_master" % k])
        np.testing.assert_array_equal(poses, g["step%d_poses" % k])
        np.testing.assert_array_equal(wts, g["step%d_weights" % k])
        np.testing.assert_array_equal(m.payload, g["step%d_payload" % k])
        np.testing.assert_array_equal(aux, g["step%d_aux" % k])


@pytest.mark.parametrize("name", ["mean", "affine", "tbm", "gmapping"])
def test_append_scan_with_per_point_quality_vs_reference(oracle, name):
    """AngleHistogramResiprocalOMQE (grid_map_scan_adders.h:32-43, init_occupancy_mapping.h:64-80: the `ahr` observation
    quality estimator): the per-point qualities and the maps after each scan against the compiled reference
    (tests/golden/make_golden_omqe.py), bit for bit."""
    from pyoracle_mapupdate import append_scan_q, omqe_quality
    g = load("map_update_ahr.npz")
    m, aux, rule = fresh_map(g, name)
    lo, hi = g["crop"]
    for k in range(int(g["n_steps"])):
        q, blur, max_range = g["step%d_params" % k]
        bq = omqe_quality(oracle, 1, g["step%d_range" % k], g["step%d_angle" % k])
        np.testing.assert_array_equal(bq, g["step%d_quality" % k])
        assert bq.min() < 0.5 * bq.max()  # the estimator really distinguishes points
        append_scan_q(oracle, m, aux, rule, g["step%d_pose" % k], g["step%d_range" % k], g["step%d_angle" % k], bq,
                      g["step%d_occ" % k], quality=q, base=g[name + "_base"], blur=blur, max_range=max_range)
        np.testing.assert_array_equal(m.payload[lo:hi, lo:hi], g["%s_step%d_payload" % (name, k)], err_msg="step %d" % k)
        if aux is not None:
            np.testing.assert_array_equal(aux[lo:hi, lo:hi], g["%s_step%d_aux" % (name, k)])
    # (idle estimator: all ones)
    assert np.array_equal(omqe_quality(oracle, 0, g["step0_range"], g["step0_angle"]), np.ones(g["step0_range"].size))


def test_cfg5_geometry_cached_provider_vs_reference(oracle):
    """cfg5's geometry (0.025 m cells, 1080 beams, walks of up to 676 cells, area estimator, blur 0.1 m) with the
    CACHED trigonometry provider (tests/golden/make_golden_cfg5_cached.py): the reference evaluates the device's own
    angle-addition form, so there is no raw-provider caveat -- every cell of every snapshot, counters and payload,
    bit for bit."""
    from helpers import dense_snapshot
    from pyoracle import TRIG_CACHED, Oracle, ScanData
    from pyoracle_mapupdate import append_scan_ex
    g = load("cfg5_cached.npz")
    w, h = [int(v) for v in g["size"]]
    tab_sin, tab_cos = Oracle().trig_table(float(g["a_min"]), float(g["a_max"]), float(g["a_inc"]))
    for hist, poses, checks in (("P", g["poses_p"], {0: "P0", 2: "P2"}), ("Q", g["poses_q"], {2: "Q2"})):
        pay = np.tile(g["unknown"][:3], (h, w, 1)).astype(np.float64)
        m = GridMapData(CELL_GMAPPING, pay, g["origin"], float(g["scale"]), g["unknown"][:3])
        aux = np.zeros((h, w, 2))
        for k in range(3):
            rng, ang = g["scan%d_range" % k], g["angle"][g["scan%d_beam" % k]]
            tr = ScanData(rng, ang, None, None, TRIG_CACHED, float(g["a_min"]), float(g["a_inc"]), tab_sin, tab_cos)
            append_scan_ex(oracle, m, aux, RULE_GMAPPING, poses[k], rng, ang, None, base=g["base"], blur=float(g["blur"]),
                           est_kind=1, shift_amount=float(g["shift_amount"]), trig=tr)
            if k in checks:
                want_p, want_a = dense_snapshot(g, checks[k])
                np.testing.assert_array_equal(aux, want_a, err_msg=checks[k])
                np.testing.assert_array_equal(m.payload, want_p, err_msg=checks[k])
