"""Shared test helpers: golden-fixture loading (tests/golden/*.npz, made by make_golden.py)."""
import os

import numpy as np
from pyoracle import (CELL_GMAPPING, CELL_OCC, CELL_TBM, TRIG_CACHED, TRIG_RAW, GridMapData,
                      ScanData)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SCENES = ["mean_raw", "mean_cached", "tbm_raw", "tbm_cached", "affine_raw"]


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


def map_from(g, prefix="map_"):
    return GridMapData(int(g[prefix + "cell_model"]), g[prefix + "payload"], g[prefix + "origin"],
                       float(g[prefix + "scale"]), g[prefix + "unknown"],
                       bool(int(g[prefix + "bounded"])))


def trig_scan(g, rng, ang, weight=None, factor=None):
    """ScanData carrying the scene's trig provider."""
    if "trig_mode" in g and int(g["trig_mode"]) == TRIG_CACHED:
        return ScanData(rng, ang, weight, factor, TRIG_CACHED, float(g["a_min"]),
                        float(g["a_inc"]), g["tab_sin"], g["tab_cos"])
    return ScanData(rng, ang, weight, factor, TRIG_RAW)


def filtered_scan(g):
    return trig_scan(g, g["f_range"], g["f_angle"], g["f_weight"], g["f_factor"])


def trace(g, prefix):
    return dict(prob=float(g[prefix + "prob"]), delta=g[prefix + "delta"],
                n_calls=int(g[prefix + "n_calls"]), poses=g[prefix + "poses"],
                scores=g[prefix + "scores"], accepted=g[prefix + "accepted"])


def assert_trace_equal(t, ref, exact_scores=True, rtol=1e-12):
    assert t["n_calls"] == ref["n_calls"]
    np.testing.assert_array_equal(t["accepted"], ref["accepted"])
    np.testing.assert_array_equal(t["poses"], ref["poses"])
    if exact_scores:
        np.testing.assert_array_equal(t["scores"], ref["scores"])
        assert t["prob"] == ref["prob"]
    else:
        np.testing.assert_allclose(t["scores"], ref["scores"], rtol=rtol, atol=0)
        np.testing.assert_allclose(t["prob"], ref["prob"], rtol=rtol, atol=0)
    np.testing.assert_array_equal(t["delta"], ref["delta"])


def dense_snapshot(g, name):
    """(payload[h, w, 3], counters[h, w, 2]) of a sparsely stored GMapping map snapshot (tests/golden/cfg5_cached.npz:
    the cells that differ from the never-observed prototype, as flat index deltas + rows)."""
    w, h = [int(v) for v in g["size"]]
    pay = np.tile(np.asarray(g["unknown"], dtype=np.float64)[:3], (h, w, 1))
    aux = np.zeros((h, w, 2))
    flat = np.cumsum(g[name + "_idx_delta"].astype(np.int64))
    pay.reshape(-1, 3)[flat] = g[name + "_payload"]
    aux.reshape(-1, 2)[flat] = g[name + "_counters"]
    return pay, aux
