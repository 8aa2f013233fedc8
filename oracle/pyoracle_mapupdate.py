"""ctypes binding of orc_append_scan (oracle/map_update_oracle.c) -- TEST INFRASTRUCTURE ONLY."""
import ctypes as C

import numpy as np
from pyoracle import (OrcScan, _d, _dp, _i, _ip, _map_struct, _scan_struct, f64, i32, ScanData)

RULE_LAST, RULE_AFFINE, RULE_MEAN, RULE_TBM, RULE_GMAPPING = range(5)
AUX_STRIDE = {RULE_MEAN: 1, RULE_GMAPPING: 2}


def append_scan(oracle, gmap, aux, rule, pose, rng, ang, is_occ=None, quality=1.0,
                base=(0.95, 1.0, 0.01, 1.0), blur=0.0, max_range=float("inf"), trig=None):
    """In-place GridMapScanAdder::append_scan on gmap.payload (and aux).  Returns #cell updates."""
    L = oracle.lib
    L.orc_append_scan.restype = C.c_longlong
    L.orc_append_scan.argtypes = [C.c_void_p, _dp, _dp, C.c_int, _dp, C.c_int, _dp, _dp, _ip,
                                  C.c_void_p, C.c_double, _dp, C.c_double, C.c_double]
    rng, ang, pose, b = f64(rng), f64(ang), f64(pose), f64(base)
    occ = i32(is_occ) if is_occ is not None else np.ones(rng.size, np.int32)
    m = _map_struct(gmap)
    ts = _scan_struct(trig or ScanData(rng, ang))
    assert gmap.payload.flags["C_CONTIGUOUS"] and gmap.payload.flags["WRITEABLE"]
    res = L.orc_append_scan(C.byref(m), _d(gmap.payload), _d(aux) if aux is not None else None, rule,
                            _d(pose), rng.size, _d(rng), _d(ang), _i(occ), C.byref(ts), quality, _d(b),
                            blur, max_range)
    if res < 0:
        raise ValueError("a touched cell lies outside the map window")
    return int(res)


def omqe_quality(oracle, kind, rng, ang):
    """ObservationMappingQualityEstimator::quality of every point: kind 0 idle, 1 angle-histogram reciprocal."""
    L = oracle.lib
    L.orc_omqe_quality.restype = None
    L.orc_omqe_quality.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp]
    rng, ang = f64(rng), f64(ang)
    out = np.zeros(rng.size)
    L.orc_omqe_quality(kind, rng.size, _d(rng), _d(ang), _d(out))
    return out


def append_scan_q(oracle, gmap, aux, rule, pose, rng, ang, beam_quality, is_occ=None, quality=1.0,
                  base=(0.95, 1.0, 0.01, 1.0), blur=0.0, max_range=float("inf"), trig=None, est_kind=0,
                  shift_amount=0.0):
    """append_scan with a per-point observation quality (the OMQE's values)."""
    L = oracle.lib
    L.orc_append_scan_q.restype = C.c_longlong
    L.orc_append_scan_q.argtypes = [C.c_void_p, _dp, _dp, C.c_int, _dp, C.c_int, _dp, _dp, _ip, C.c_void_p, C.c_double,
                                    _dp, C.c_double, C.c_double, C.c_int, C.c_double, _dp]
    rng, ang, pose, b, bq = f64(rng), f64(ang), f64(pose), f64(base), f64(beam_quality)
    occ = i32(is_occ) if is_occ is not None else np.ones(rng.size, np.int32)
    m = _map_struct(gmap)
    ts = _scan_struct(trig or ScanData(rng, ang))
    res = L.orc_append_scan_q(C.byref(m), _d(gmap.payload), _d(aux) if aux is not None else None, rule, _d(pose),
                              rng.size, _d(rng), _d(ang), _i(occ), C.byref(ts), quality, _d(b), blur, max_range,
                              est_kind, shift_amount, _d(bq))
    if res < 0:
        raise ValueError("a touched cell lies outside the map window")
    return int(res)


def append_scan_ex(oracle, gmap, aux, rule, pose, rng, ang, is_occ=None, quality=1.0,
                   base=(0.95, 1.0, 0.01, 1.0), blur=0.0, max_range=float("inf"), trig=None,
                   est_kind=0, shift_amount=0.0):
    """append_scan with the const (0) or the area (1) occupancy estimator."""
    L = oracle.lib
    L.orc_append_scan_ex.restype = C.c_longlong
    L.orc_append_scan_ex.argtypes = [C.c_void_p, _dp, _dp, C.c_int, _dp, C.c_int, _dp, _dp, _ip,
                                     C.c_void_p, C.c_double, _dp, C.c_double, C.c_double, C.c_int,
                                     C.c_double]
    rng, ang, pose, b = f64(rng), f64(ang), f64(pose), f64(base)
    occ = i32(is_occ) if is_occ is not None else np.ones(rng.size, np.int32)
    m = _map_struct(gmap)
    ts = _scan_struct(trig or ScanData(rng, ang))
    res = L.orc_append_scan_ex(C.byref(m), _d(gmap.payload), _d(aux) if aux is not None else None, rule,
                               _d(pose), rng.size, _d(rng), _d(ang), _i(occ), C.byref(ts), quality, _d(b),
                               blur, max_range, est_kind, shift_amount)
    if res < 0:
        raise ValueError("a touched cell lies outside the map window")
    return int(res)


class OrcAdder(C.Structure):
    _fields_ = [("base4", C.c_double * 4), ("blur", C.c_double), ("max_range", C.c_double),
                ("est_kind", C.c_int), ("shift_amount", C.c_double)]


def gmapping_enable_update(oracle, pf, gmap, aux, base=(0.95, 1.0, 0.01, 1.0), blur=0.0,
                           max_range=float("inf"), est_kind=0, shift_amount=0.0):
    """Switch on the map update inside OrcGmappingHandle.step: every matching particle appends its
    scan to gmap.payload / aux (the shared map) before the next particle matches."""
    L = oracle.lib
    L.orc_gmapping_set_update.argtypes = [C.c_void_p, C.c_void_p, _dp, _dp]
    a = OrcAdder()
    for k in range(4):
        a.base4[k] = float(base[k])
    a.blur, a.max_range, a.est_kind, a.shift_amount = blur, max_range, est_kind, shift_amount
    L.orc_gmapping_set_update(pf.h, C.byref(a), _d(gmap.payload), _d(aux))


def gmapping_enable_particle_maps(oracle, pf, gmap, aux=None, base=(0.95, 1.0, 0.01, 1.0), blur=0.0,
                                  max_range=float("inf"), est_kind=0, shift_amount=0.0):
    """Every particle of the OrcGmappingHandle gets its own copy of gmap.payload (+ aux: hits, tries);
    the step's map argument keeps describing the geometry."""
    L = oracle.lib
    L.orc_gmapping_set_particle_maps.argtypes = [C.c_void_p, C.c_void_p, _dp, C.c_size_t, _dp, C.c_size_t]
    a = OrcAdder()
    for k in range(4):
        a.base4[k] = float(base[k])
    a.blur, a.max_range, a.est_kind, a.shift_amount = blur, max_range, est_kind, shift_amount
    pay = f64(gmap.payload)
    n_cells = pay.shape[0] * pay.shape[1]
    ax = f64(aux) if aux is not None else None
    L.orc_gmapping_set_particle_maps(pf.h, C.byref(a), _d(pay), pay.size, _d(ax) if ax is not None else None,
                                     2 * n_cells)
    pf._pm_shape = pay.shape


def gmapping_particle_map(oracle, pf, particle):
    """(payload[h, w, 3], counters[h, w, 2]) of one particle's own map."""
    L = oracle.lib
    L.orc_gmapping_copy_particle_map.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
    h, w, st = pf._pm_shape
    pay, aux = np.zeros((h, w, st)), np.zeros((h, w, 2))
    L.orc_gmapping_copy_particle_map(pf.h, particle, _d(pay), _d(aux))
    return pay, aux


def gmapping_particle_map_append(oracle, pf, gmap, particle, pose, rng, ang, is_occ=None, trig=None):
    """GridMapScanAdder::append_scan on ONE particle's own map (adder parameters of gmapping_enable_particle_maps),
    from `pose`, with the scan's trig provider `trig` (None = raw).  Returns #cell updates."""
    L = oracle.lib
    L.orc_gmapping_particle_map_append.restype = C.c_longlong
    L.orc_gmapping_particle_map_append.argtypes = [C.c_void_p, C.c_void_p, C.c_int, _dp, C.c_int, _dp, _dp, _ip,
                                                   C.c_void_p]
    rng, ang, pose = f64(rng), f64(ang), f64(pose)
    occ = i32(is_occ) if is_occ is not None else np.ones(rng.size, np.int32)
    m = _map_struct(gmap)
    ts = _scan_struct(trig or ScanData(rng, ang))
    res = L.orc_gmapping_particle_map_append(pf.h, C.byref(m), particle, _d(pose), rng.size, _d(rng), _d(ang), _i(occ),
                                             C.byref(ts))
    if res < 0:
        raise ValueError("particle map append failed (%d)" % res)
    return int(res)
