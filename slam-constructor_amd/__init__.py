"""slam-constructor_amd -- MI355X-native scan-matching / particle-likelihood engine.

Thin ctypes binding over the C-ABI shared library (include/slamhip.h, built in-tree as
slam-constructor_amd/libslamhip.so from csrc/).  The directory name carries a hyphen, so load it
with ``load_package()`` from ``__graft_entry__`` (module name ``slam_constructor_amd``).

There is NO CPU fallback in here: if the HIP library is missing or no GPU is usable, ``load()``
/ ``Context()`` raise.  The CPU oracle under oracle/ is test infrastructure and is never imported
from this package.
"""
import atexit
import ctypes as C
import os
import subprocess
import weakref

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG_DIR, "libslamhip.so")
# the same library built with -DSLAMHIP_TESTING: the debugging / fault-injection hooks the shipped one does not export
TESTING_LIB_PATH = os.path.join(PKG_DIR, "libslamhip_testing.so")
CSRC = os.path.join(PKG_DIR, "csrc")

CELL_OCC, CELL_TBM, CELL_GMAPPING = 0, 1, 2
OOPE_OBSTACLE, OOPE_MAX, OOPE_MEAN, OOPE_OVERLAP, OOPE_GMAPPING = range(5)
OIE_DISCREPANCY, OIE_OCCUPANCY = 0, 1
SUM_TREE256, SUM_SEQUENTIAL = 0, 1
POSE_TRIG_DEVICE, POSE_TRIG_HOST, POSE_TRIG_RAW_EXACT = 0, 1, 2
(OPT_LOW_LATENCY, OPT_STAGE_POSES, OPT_FILTER_CHAINS, OPT_K6_PATH, OPT_K6_BATCH_FAST, OPT_K6_BATCH_KEY64,
 OPT_RESIDENT_CHAINS, OPT_TBM_PLANE, OPT_INERT_TAIL) = range(9)
TRIG_RAW, TRIG_CACHED = 0, 1
STRIDE = {CELL_OCC: 1, CELL_TBM: 4, CELL_GMAPPING: 3}

EXPORTS = """slamhip_last_error slamhip_device_count slamhip_ctx_create slamhip_ctx_destroy
slamhip_ctx_synchronize slamhip_ctx_stream slamhip_ctx_set_option slamhip_ctx_get_option slamhip_map_bind slamhip_map_upload_window
slamhip_map_apply_dirty slamhip_map_release slamhip_map_download_window slamhip_scan_upload
slamhip_beam_trig_raw slamhip_beam_trig_cached slamhip_filter_scan slamhip_scan_weights
slamhip_score_poses slamhip_score_poses_device slamhip_gm_cache_reset slamhip_gm_cache_get slamhip_gm_cache_set
slamhip_profile_enable slamhip_profile_read slamhip_profile_read_map_update slamhip_matcher_create_mc slamhip_matcher_create_hc
slamhip_matcher_create_bf slamhip_matcher_destroy slamhip_matcher_reset_state
slamhip_map_set_auto_grow slamhip_map_info slamhip_map_set_deferred slamhip_map_drain slamhip_matcher_set_observer slamhip_matcher_set_batch slamhip_matcher_set_device_chain slamhip_matcher_set_tie_check slamhip_matcher_process_scan
slamhip_matcher_process_raw_scan slamhip_matcher_stats slamhip_matcher_chain_stats slamhip_matcher_tail_stats slamhip_matcher_resident_stats slamhip_matcher_timing slamhip_pf_normalize slamhip_pf_resampling_is_required slamhip_pf_resample
slamhip_pf_heaviest slamhip_gmapping_create slamhip_gmapping_destroy slamhip_gmapping_predict_match
slamhip_gmapping_plan_resample slamhip_gmapping_blob_size slamhip_gmapping_export
slamhip_gmapping_import slamhip_gmapping_step slamhip_gmapping_set slamhip_gmapping_get
slamhip_gmapping_stats slamhip_gmapping_set_map_update slamhip_map_append_scan slamhip_map_download_aux
slamhip_gmapping_enable_particle_maps slamhip_gmapping_particle_map_download
slamhip_gmapping_particle_map_stats slamhip_gmapping_particle_map_export_size
slamhip_gmapping_particle_map_export slamhip_gmapping_import_maps slamhip_gmapping_particle_maps_append
slamhip_shard_unique_id slamhip_shard_init slamhip_shard_destroy slamhip_shard_info slamhip_shard_allgather
slamhip_shard_stats slamhip_gmapping_set_shard_chain slamhip_gmapping_match_begin slamhip_gmapping_carry_record
slamhip_gmapping_carry_fix slamhip_gmapping_carry_commit slamhip_gmapping_match_finish
slamhip_gmapping_step_sharded slamhip_matcher_process_scan_batch slamhip_matcher_batch_stats slamhip_scan_store slamhip_scan_select
slamhip_shard_attach slamhip_shard_exchange slamhip_shard_p2p_stats slamhip_shard_set_timeout slamhip_gmapping_match_abort
slamhip_gmapping_migration_stats slamhip_map_append_scan_q slamhip_omqe_quality slamhip_scan_filter_upload
slamhip_scan_set_angles slamhip_libm_variant slamhip_libm_eval slamhip_map_append_scan_raw""".split()

SHARD_ID_BYTES = 128

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class SlamHipError(RuntimeError):
    pass


class ShardMsg(C.Structure):
    """slamhip_shard_msg"""
    _fields_ = [("peer", C.c_int), ("buf", C.c_void_p), ("bytes", C.c_size_t)]


class MatchJob(C.Structure):
    """slamhip_match_job: one match of slamhip_matcher_process_scan_batch."""
    _fields_ = [("map_id", C.c_int), ("scan_slot", C.c_int), ("n", C.c_int), ("range", _dp), ("cos_a", _dp), ("sin_a", _dp),
                ("weight", _dp), ("factor", _dp), ("init_pose", C.c_double * 3)]


class CarryRecord(C.Structure):
    _fields_ = [("has_active", C.c_int), ("first_cx", C.c_int), ("first_cy", C.c_int), ("first_v0", C.c_double),
                ("carry_cx", C.c_int), ("carry_cy", C.c_int), ("carry_prob", C.c_double)]


class SpeCfg(C.Structure):
    _fields_ = [("oope", C.c_int), ("oie", C.c_int), ("area", C.c_double * 4),
                ("gm_fullness_th", C.c_double), ("gm_window", C.c_int), ("sum_order", C.c_int),
                ("pose_trig", C.c_int)]


RULE_LAST, RULE_AFFINE, RULE_MEAN, RULE_TBM, RULE_GMAPPING = range(5)


class ScanAdderCfg(C.Structure):
    _fields_ = [("rule", C.c_int), ("scan_quality", C.c_double), ("base_occupied_prob", C.c_double),
                ("base_occupied_qual", C.c_double), ("base_empty_prob", C.c_double),
                ("base_empty_qual", C.c_double), ("blur", C.c_double), ("max_range", C.c_double),
                ("occupancy_estimator", C.c_int), ("area_shift_amount", C.c_double)]


class GmappingParams(C.Structure):
    _fields_ = [("mean_sample_xy", C.c_double), ("sigma_sample_xy", C.c_double),
                ("mean_sample_th", C.c_double), ("sigma_sample_th", C.c_double),
                ("min_sm_lim_xy", C.c_double), ("max_sm_lim_xy", C.c_double),
                ("min_sm_lim_th", C.c_double), ("max_sm_lim_th", C.c_double),
                ("hc_failed_rounds_limit", C.c_uint), ("hc_translation", C.c_double),
                ("hc_rotation", C.c_double), ("sp_skip_rate", C.c_uint),
                ("sp_max_usable_range", C.c_double), ("oope_fullness_th", C.c_double),
                ("oope_window", C.c_int), ("pose_trig", C.c_int)]


def gmapping_params(gp8=(0.0, 0.1, 0.0, 0.03, 0.6, 0.8, 0.3, 0.4), hc=(6, 0.1, 0.1), skip_rate=0,
                    max_range=-1.0, fullness_th=0.1, window=1, pose_trig=POSE_TRIG_DEVICE):
    """Defaults of init_gmapping (src/slams/gmapping/init_gmapping.h:15-60)."""
    p = GmappingParams()
    (p.mean_sample_xy, p.sigma_sample_xy, p.mean_sample_th, p.sigma_sample_th, p.min_sm_lim_xy,
     p.max_sm_lim_xy, p.min_sm_lim_th, p.max_sm_lim_th) = [float(v) for v in gp8]
    p.hc_failed_rounds_limit, p.hc_translation, p.hc_rotation = int(hc[0]), hc[1], hc[2]
    p.sp_skip_rate, p.sp_max_usable_range = skip_rate, max_range
    p.oope_fullness_th, p.oope_window, p.pose_trig = fullness_th, window, pose_trig
    return p


OBS_FN = C.CFUNCTYPE(None, C.c_void_p, _dp, C.c_double)


class Observer(C.Structure):
    _fields_ = [("user", C.c_void_p), ("on_scan_test", OBS_FN), ("on_pose_update", OBS_FN),
                ("on_matching_end", OBS_FN)]


def spe_cfg(oope=OOPE_OBSTACLE, oie=OIE_DISCREPANCY, area=(0, 0, 0, 0), gm_th=0.1, gm_window=1,
            sum_order=SUM_TREE256, pose_trig=POSE_TRIG_DEVICE):
    c = SpeCfg()
    c.oope, c.oie = oope, oie
    for k in range(4):
        c.area[k] = float(area[k])
    c.gm_fullness_th, c.gm_window = gm_th, gm_window
    c.sum_order, c.pose_trig = sum_order, pose_trig
    return c


def build(verbose=False):
    """Compile the HIP kernels + C-ABI for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-j8", "-C", CSRC], stdout=out)
    return LIB_PATH


_libs = {}


def load(testing=False):
    """dlopen libslamhip.so and declare the prototypes.  Raises if it is missing: the product
    path has no fallback.  testing=True: libslamhip_testing.so -- the same sources with the test hooks compiled in
    (slamhip_*debug*), for the tests that need one; objects of the two libraries do not mix (pass testing=True to
    Context and take matchers / filters from that context)."""
    if testing in _libs:
        return _libs[testing]
    path = TESTING_LIB_PATH if testing else LIB_PATH
    if not os.path.exists(path):
        raise SlamHipError("%s is not built (run __graft_entry__.build()); "
                           "there is no CPU fallback" % os.path.basename(path))
    L = C.CDLL(path)
    vp, d, i, u = C.c_void_p, C.c_double, C.c_int, C.c_uint
    L.slamhip_last_error.restype = C.c_char_p
    L.slamhip_ctx_create.argtypes = [i, C.POINTER(vp)]
    L.slamhip_ctx_destroy.argtypes = [vp]
    L.slamhip_ctx_synchronize.argtypes = [vp]
    L.slamhip_ctx_stream.restype = vp
    L.slamhip_ctx_stream.argtypes = [vp]
    L.slamhip_map_bind.argtypes = [vp, i, i, i, i, i, i, d, _dp]
    L.slamhip_map_upload_window.argtypes = [vp, i, i, i, i, i, _dp]
    L.slamhip_map_download_window.argtypes = [vp, i, i, i, i, i, _dp]
    L.slamhip_map_apply_dirty.argtypes = [vp, i, i, _ip, _dp]
    L.slamhip_map_release.argtypes = [vp, i]
    L.slamhip_map_set_auto_grow.argtypes = [vp, i, i]
    L.slamhip_map_info.argtypes = [vp, i, _ip, _ip, _ip, _ip, _ip, _dp, C.POINTER(C.c_longlong)]
    L.slamhip_map_set_deferred.argtypes = [vp, i]
    L.slamhip_map_drain.argtypes = [vp, C.POINTER(C.c_longlong)]
    L.slamhip_scan_upload.argtypes = [vp, i, _dp, _dp, _dp, _dp, _dp]
    L.slamhip_scan_store.argtypes = [vp, i, i, _dp, _dp, _dp, _dp, _dp]
    L.slamhip_scan_select.argtypes = [vp, i]
    L.slamhip_map_append_scan.argtypes = [vp, i, C.POINTER(ScanAdderCfg), _dp, i, _dp, _dp, _dp, _ip,
                                          C.POINTER(C.c_longlong)]
    L.slamhip_map_download_aux.argtypes = [vp, i, i, i, i, i, _dp]
    L.slamhip_beam_trig_raw.argtypes = [i, _dp, _dp, _dp]
    L.slamhip_beam_trig_cached.argtypes = [i, _dp, d, d, d, _dp, _dp]
    L.slamhip_filter_scan.argtypes = [i, _dp, _dp, _ip, i, d, d, i, _dp, _dp, _dp, u, d, i, i, i,
                                      i, i, d, _ip, _ip]
    L.slamhip_scan_weights.argtypes = [i, i, _dp, _dp, _dp]
    L.slamhip_score_poses.argtypes = [vp, i, C.POINTER(SpeCfg), i, _dp, _dp]
    L.slamhip_score_poses_device.argtypes = [vp, i, C.POINTER(SpeCfg), i, vp, vp]
    L.slamhip_gm_cache_reset.argtypes = [vp]
    L.slamhip_gm_cache_get.argtypes = [vp, _ip, _dp]
    L.slamhip_gm_cache_set.argtypes = [vp, _ip, d]
    L.slamhip_profile_enable.argtypes = [vp, i]
    L.slamhip_profile_read.argtypes = [vp, _dp, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), i]
    L.slamhip_profile_read_map_update.argtypes = [vp, _dp, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), i]
    L.slamhip_matcher_create_mc.argtypes = [vp, C.POINTER(SpeCfg), u, d, d, u, u, C.POINTER(vp)]
    L.slamhip_matcher_create_hc.argtypes = [vp, C.POINTER(SpeCfg), u, d, d, C.POINTER(vp)]
    L.slamhip_matcher_create_bf.argtypes = [vp, C.POINTER(SpeCfg), _dp, C.POINTER(vp)]
    L.slamhip_matcher_destroy.argtypes = [vp]
    L.slamhip_matcher_reset_state.argtypes = [vp]
    L.slamhip_matcher_set_observer.argtypes = [vp, C.POINTER(Observer)]
    L.slamhip_matcher_set_batch.argtypes = [vp, i]
    L.slamhip_matcher_set_device_chain.argtypes = [vp, i, i]
    L.slamhip_matcher_set_tie_check.argtypes = [vp, i]
    L.slamhip_matcher_process_scan.argtypes = [vp, i, _dp, _dp, _dp]
    L.slamhip_matcher_stats.argtypes = [vp] + [C.POINTER(C.c_longlong)] * 3
    L.slamhip_matcher_timing.argtypes = [vp, _dp, _dp, _dp, _dp]
    L.slamhip_matcher_chain_stats.argtypes = [vp, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
    L.slamhip_matcher_tail_stats.argtypes = [vp, C.POINTER(C.c_longlong)]
    L.slamhip_matcher_resident_stats.argtypes = [vp, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
    L.slamhip_pf_normalize.argtypes = [i, _dp]
    L.slamhip_pf_resampling_is_required.argtypes = [i, _dp, _ip]
    L.slamhip_pf_resample.argtypes = [i, _dp, C.c_uint32, C.POINTER(C.c_uint)]
    L.slamhip_pf_heaviest.argtypes = [i, _dp, _ip]
    up = C.POINTER(C.c_uint)
    ll = C.POINTER(C.c_longlong)
    L.slamhip_gmapping_create.argtypes = [vp, C.POINTER(GmappingParams), i, i, i,
                                          C.POINTER(C.c_uint32), C.POINTER(vp)]
    L.slamhip_gmapping_destroy.argtypes = [vp]
    L.slamhip_gmapping_predict_match.argtypes = [vp, i, i, _dp, _dp, _ip, _dp, _dp]
    L.slamhip_gmapping_plan_resample.argtypes = [vp, _dp, C.c_uint32, _ip, up]
    L.slamhip_gmapping_blob_size.restype = C.c_size_t
    L.slamhip_gmapping_export.argtypes = [vp, vp]
    L.slamhip_gmapping_import.argtypes = [vp, vp, up]
    L.slamhip_gmapping_step.argtypes = [vp, i, i, _dp, _dp, _ip, _dp, C.c_uint32, _ip, up]
    L.slamhip_gmapping_set.argtypes = [vp, _dp, _dp]
    L.slamhip_gmapping_set_map_update.argtypes = [vp, C.POINTER(ScanAdderCfg)]
    L.slamhip_gmapping_get.argtypes = [vp, _dp, _dp, _ip]
    L.slamhip_gmapping_stats.argtypes = [vp, ll, ll, ll, ll]
    L.slamhip_gmapping_enable_particle_maps.argtypes = [vp, i, C.POINTER(ScanAdderCfg), i, i]
    L.slamhip_gmapping_particle_map_download.argtypes = [vp, i, i, i, i, i, _dp, _dp]
    L.slamhip_gmapping_particle_map_stats.argtypes = [vp, ll, ll, ll, ll, ll]
    L.slamhip_gmapping_particle_map_export_size.argtypes = [vp, i, C.POINTER(C.c_size_t)]
    L.slamhip_gmapping_particle_map_export.argtypes = [vp, i, vp, C.c_size_t]
    L.slamhip_gmapping_import_maps.argtypes = [vp, vp, up, i, _ip, C.POINTER(vp)]
    L.slamhip_gmapping_particle_maps_append.argtypes = [vp, i, _ip, _dp, i, _dp, _dp, _ip, ll]
    L.slamhip_matcher_process_scan_batch.argtypes = [vp, i, C.POINTER(MatchJob), _dp, _dp]
    L.slamhip_matcher_batch_stats.argtypes = [vp, i, ll, ll, ll, _ip]
    L.slamhip_shard_unique_id.argtypes = [vp]
    L.slamhip_shard_init.argtypes = [vp, i, i, vp]
    L.slamhip_shard_destroy.argtypes = [vp]
    L.slamhip_shard_info.argtypes = [vp, _ip, _ip]
    L.slamhip_shard_allgather.argtypes = [vp, vp, _ip, i, vp]
    L.slamhip_map_append_scan_q.argtypes = [vp, i, C.POINTER(ScanAdderCfg), _dp, i, _dp, _dp, _dp, _ip, _dp, ll]
    L.slamhip_omqe_quality.argtypes = [i, i, _dp, _dp, _dp]
    L.slamhip_shard_attach.argtypes = [vp, i, i, vp]
    L.slamhip_shard_exchange.argtypes = [vp, i, vp, i, vp]
    L.slamhip_shard_p2p_stats.argtypes = [vp, ll, ll]
    L.slamhip_gmapping_match_abort.argtypes = [vp]
    L.slamhip_gmapping_migration_stats.argtypes = [vp, ll, ll]
    L.slamhip_shard_stats.argtypes = [vp, ll, ll]
    L.slamhip_gmapping_set_shard_chain.argtypes = [vp, i]
    L.slamhip_gmapping_match_begin.argtypes = [vp, i, i, _dp, _dp, _ip, _dp]
    L.slamhip_gmapping_carry_record.argtypes = [vp, C.POINTER(CarryRecord)]
    L.slamhip_gmapping_carry_fix.argtypes = [vp, C.POINTER(CarryRecord), i, i, _ip]
    L.slamhip_gmapping_carry_commit.argtypes = [vp, C.POINTER(CarryRecord), i]
    L.slamhip_gmapping_match_finish.argtypes = [vp, _dp]
    L.slamhip_gmapping_step_sharded.argtypes = [vp, i, i, _dp, _dp, _ip, _dp, C.c_uint32, _ip, up]
    _libs[testing] = L
    return L


def omqe_quality(kind, rng, ang):
    """ObservationMappingQualityEstimator::quality per point: kind 0 idle, 1 angle-histogram reciprocal ("ahr")."""
    rng, ang = _f64(rng), _f64(ang)
    out = np.zeros(rng.size)
    _check(load().slamhip_omqe_quality(int(kind), rng.size, _d(rng), _d(ang), _d(out)))
    return out


def shard_unique_id():
    """128 bytes that rank 0 creates and every rank hands to Context.shard_init (ncclGetUniqueId)."""
    buf = np.zeros(SHARD_ID_BYTES, np.uint8)
    _check(load().slamhip_shard_unique_id(buf.ctypes.data_as(C.c_void_p)))
    return buf


# Objects still alive when the interpreter shuts down (a failed test, a script without close()) must be
# destroyed BEFORE the HIP runtime's own static destructors run: freeing device memory after that aborts the
# process at exit.  Filters and matchers first (they point into their context), contexts last.
_live = {"dep": weakref.WeakSet(), "ctx": weakref.WeakSet()}


def _close_all():
    for kind in ("dep", "ctx"):
        for obj in list(_live[kind]):
            try:
                obj.close()
            except Exception:
                pass


atexit.register(_close_all)


def _check(rc):
    if rc != 0:
        # (the message is thread-local state of whichever of the two libraries the failing call went into)
        msgs = [L_.slamhip_last_error().decode() for L_ in _libs.values()]
        raise SlamHipError("slamhip error %d: %s" % (rc, "; ".join(m for m in msgs if m) or "(no message)"))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _d(a):
    return a.ctypes.data_as(_dp)


# ---- host-side pieces of the path (no GPU needed) ------------------------------------------
def beam_trig(angle, trig_mode=TRIG_RAW, a_min=0.0, a_max=0.0, a_inc=1.0):
    """(cos a_i, sin a_i) as the scan's TrigonometryProvider tabulates them."""
    L = load()
    angle = _f64(angle)
    c, s = np.zeros(angle.size), np.zeros(angle.size)
    if trig_mode == TRIG_CACHED:
        _check(L.slamhip_beam_trig_cached(angle.size, _d(angle), a_min, a_max, a_inc, _d(c), _d(s)))
    else:
        _check(L.slamhip_beam_trig_raw(angle.size, _d(angle), _d(c), _d(s)))
    return c, s


def filter_scan(rng, ang, is_occ, pose, geom, skip_rate=0, max_range=-1.0, trig_mode=TRIG_RAW,
                a_min=0.0, a_delta=1.0, tab_sin=None, tab_cos=None):
    """WeightedMeanPointProbabilitySPE::filter_scan; geom = dict(width, height, origin, scale,
    bounded).  Returns the kept raw indices."""
    L = load()
    rng, ang, pose = _f64(rng), _f64(ang), _f64(pose)
    occ = np.ascontiguousarray(is_occ if is_occ is not None else np.ones(rng.size), dtype=np.int32)
    ts = _f64(tab_sin) if tab_sin is not None else np.zeros(1)
    tc = _f64(tab_cos) if tab_cos is not None else np.zeros(1)
    kept = np.zeros(max(rng.size, 1), np.int32)
    n = C.c_int(0)
    _check(L.slamhip_filter_scan(rng.size, _d(rng), _d(ang), occ.ctypes.data_as(_ip), trig_mode,
                                 a_min, a_delta, ts.size, _d(ts), _d(tc), _d(pose), skip_rate,
                                 max_range, int(bool(geom.get("bounded", False))), geom["width"],
                                 geom["height"], geom["origin"][0], geom["origin"][1],
                                 geom["scale"], kept.ctypes.data_as(_ip), C.byref(n)))
    return kept[:n.value].copy()


def libm_variant():
    """1: the host's libm runs glibc's FMA build of sin / cos / exp, 0: the plain build, -1: neither (no exact modes)"""
    v = C.c_int(-2)
    _check(load().slamhip_libm_variant(C.byref(v)))
    return v.value


def scan_weights(kind, rng, ang):
    L = load()
    rng, ang = _f64(rng), _f64(ang)
    out = np.zeros(rng.size)
    _check(L.slamhip_scan_weights({"even": 0, "viny": 1, "ahr": 2}[kind], rng.size, _d(rng),
                                  _d(ang), _d(out)))
    return out


def pf_normalize(w):
    w = _f64(w).copy()
    _check(load().slamhip_pf_normalize(w.size, _d(w)))
    return w


def pf_resampling_is_required(w):
    w = _f64(w)
    r = C.c_int(0)
    _check(load().slamhip_pf_resampling_is_required(w.size, _d(w), C.byref(r)))
    return bool(r.value)


def pf_resample(w, seed):
    w = _f64(w)
    out = np.zeros(w.size, np.uint32)
    _check(load().slamhip_pf_resample(w.size, _d(w), seed, out.ctypes.data_as(C.POINTER(C.c_uint))))
    return out


def pf_heaviest(w):
    w = _f64(w)
    r = C.c_int(0)
    _check(load().slamhip_pf_heaviest(w.size, _d(w), C.byref(r)))
    return r.value


# ---- GPU objects ---------------------------------------------------------------------------------
class Context:
    """One GPU + one HIP stream (slamhip_ctx)."""

    def __init__(self, device=0, testing=False):
        self.L = load(testing)
        h = C.c_void_p()
        _check(self.L.slamhip_ctx_create(device, C.byref(h)))
        self.h = h
        self.device = device
        self._deps = weakref.WeakSet()  # matchers / filters created on this context: closed before it
        _live["ctx"].add(self)

    def close(self):
        if getattr(self, "h", None):
            for d in list(self._deps):
                d.close()
            self.L.slamhip_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        _check(self.L.slamhip_ctx_synchronize(self.h))

    def stream(self):
        return self.L.slamhip_ctx_stream(self.h)

    def set_option(self, option, value):
        """slamhip_ctx_set_option: OPT_* switches between execution paths with the same results."""
        self.L.slamhip_ctx_set_option.argtypes = [C.c_void_p, C.c_int, C.c_int]
        _check(self.L.slamhip_ctx_set_option(self.h, int(option), int(value)))

    def get_option(self, option):
        self.L.slamhip_ctx_get_option.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        v = C.c_int(0)
        _check(self.L.slamhip_ctx_get_option(self.h, int(option), C.byref(v)))
        return v.value

    # map mirror
    def map_bind(self, map_id, cell_model, width, height, origin, scale, unknown):
        unk = np.zeros(4)
        unk[:STRIDE[cell_model]] = np.asarray(unknown, dtype=np.float64).ravel()[:STRIDE[cell_model]]
        _check(self.L.slamhip_map_bind(self.h, map_id, cell_model, width, height, origin[0],
                                       origin[1], scale, _d(unk)))

    def map_upload_window(self, map_id, x0, y0, payload):
        p = _f64(payload)
        h, w = p.shape[:2]
        _check(self.L.slamhip_map_upload_window(self.h, map_id, x0, y0, w, h, _d(p)))

    def map_download_window(self, map_id, x0, y0, w, h, stride):
        out = np.zeros((h, w, stride))
        _check(self.L.slamhip_map_download_window(self.h, map_id, x0, y0, w, h, _d(out)))
        return out

    def map_apply_dirty(self, map_id, coords_xy, payloads):
        xy = np.ascontiguousarray(coords_xy, dtype=np.int32).reshape(-1, 2)
        p = _f64(payloads)
        _check(self.L.slamhip_map_apply_dirty(self.h, map_id, xy.shape[0],
                                              xy.ctypes.data_as(_ip), _d(p)))

    def map_append_scan(self, map_id, rule, pose, rng, cos_a, sin_a, is_occ=None, quality=1.0,
                        base=(0.95, 1.0, 0.01, 1.0), blur=0.0, max_range=float("inf"), estimator=0,
                        shift_amount=0.0, beam_quality=None):
        """GridMapScanAdder::append_scan on the HBM mirror (estimator 0 const, 1 area); returns the
        number of cell updates.  beam_quality: the observation quality estimator's value per point (omqe_quality)."""
        cfg = ScanAdderCfg(rule, quality, base[0], base[1], base[2], base[3], blur, max_range,
                           estimator, shift_amount)
        rng, cos_a, sin_a, pose = _f64(rng), _f64(cos_a), _f64(sin_a), _f64(pose)
        occ = np.ascontiguousarray(is_occ, dtype=np.int32) if is_occ is not None else None
        bq = _f64(beam_quality) if beam_quality is not None else None
        assert bq is None or bq.size == rng.size
        nu = C.c_longlong(0)
        _check(self.L.slamhip_map_append_scan_q(self.h, map_id, C.byref(cfg), _d(pose), rng.size, _d(rng),
                                                _d(cos_a), _d(sin_a),
                                                occ.ctypes.data_as(_ip) if occ is not None else None,
                                                _d(bq) if bq is not None else None, C.byref(nu)))
        return nu.value

    def map_append_scan_raw(self, map_id, rule, pose, rng, angle, is_occ=None, quality=1.0,
                            base=(0.95, 1.0, 0.01, 1.0), blur=0.0, max_range=float("inf"), estimator=0,
                            shift_amount=0.0, beam_quality=None):
        """append_scan with the reference's default RawTrigonometryProvider, bit for bit (slamhip_map_append_scan_raw)"""
        cfg = ScanAdderCfg(rule, quality, base[0], base[1], base[2], base[3], blur, max_range,
                           estimator, shift_amount)
        rng, angle, pose = _f64(rng), _f64(angle), _f64(pose)
        occ = np.ascontiguousarray(is_occ, dtype=np.int32) if is_occ is not None else None
        bq = _f64(beam_quality) if beam_quality is not None else None
        assert bq is None or bq.size == rng.size
        nu = C.c_longlong(0)
        self.L.slamhip_map_append_scan_raw.argtypes = [C.c_void_p, C.c_int, C.c_void_p, _dp, C.c_int, _dp, _dp, _ip, _dp,
                                                       C.POINTER(C.c_longlong)]
        _check(self.L.slamhip_map_append_scan_raw(self.h, map_id, C.byref(cfg), _d(pose), rng.size, _d(rng), _d(angle),
                                                  occ.ctypes.data_as(_ip) if occ is not None else None,
                                                  _d(bq) if bq is not None else None, C.byref(nu)))
        return nu.value

    def map_download_aux(self, map_id, x0, y0, w, h, stride):
        out = np.zeros((h, w, stride))
        _check(self.L.slamhip_map_download_aux(self.h, map_id, x0, y0, w, h, _d(out)))
        return out

    def map_release(self, map_id):
        _check(self.L.slamhip_map_release(self.h, map_id))

    def map_set_auto_grow(self, map_id, on=True):
        """An unbounded map: map_append_scan grows the window instead of failing (UnboundedPlainGridMap)."""
        _check(self.L.slamhip_map_set_auto_grow(self.h, map_id, int(bool(on))))

    def map_set_deferred(self, on=True):
        """Map updates queued, not awaited: map_append_scan returns -1 updates, map_drain() reports them."""
        _check(self.L.slamhip_map_set_deferred(self.h, 1 if on else 0))

    def map_drain(self):
        n = C.c_longlong(0)
        _check(self.L.slamhip_map_drain(self.h, C.byref(n)))
        return n.value

    def map_info(self, map_id):
        v = [C.c_int() for _ in range(5)]
        scale, grown = C.c_double(), C.c_longlong()
        _check(self.L.slamhip_map_info(self.h, map_id, *[C.byref(x) for x in v], C.byref(scale), C.byref(grown)))
        return dict(cell_model=v[0].value, width=v[1].value, height=v[2].value, origin=(v[3].value, v[4].value),
                    scale=scale.value, times_grown=grown.value)

    def upload_map(self, map_id, m):
        """m: any object with cell_model, payload[h,w,stride], origin, scale, unknown."""
        self.map_bind(map_id, m.cell_model, m.width, m.height, m.origin, m.scale, m.unknown)
        self.map_upload_window(map_id, 0, 0, m.payload)

    # scan
    def scan_upload(self, rng, cos_a, sin_a, weight, factor=None):
        rng, cos_a, sin_a, weight = _f64(rng), _f64(cos_a), _f64(sin_a), _f64(weight)
        fac = _f64(factor) if factor is not None else np.ones(rng.size)
        _check(self.L.slamhip_scan_upload(self.h, rng.size, _d(rng), _d(cos_a), _d(sin_a),
                                          _d(weight), _d(fac)))

    def make_scan_upload(self, rng, cos_a, sin_a, weight, factor=None):
        """Argument block of slamhip_scan_upload made once for a filtered scan (a caller that holds its scans in C arrays
        pays no conversions per call): returns upload()."""
        rng, cos_a, sin_a, weight = _f64(rng), _f64(cos_a), _f64(sin_a), _f64(weight)
        fac = _f64(factor) if factor is not None else np.ones(rng.size)
        fn = self.L.slamhip_scan_upload
        args = (self.h, rng.size, _d(rng), _d(cos_a), _d(sin_a), _d(weight), _d(fac))
        keep = (rng, cos_a, sin_a, weight, fac)

        def upload(_keep=keep):
            rc = fn(*args)
            if rc:
                _check(rc)
        return upload

    def make_map_append_scan(self, map_id, rule, rng, cos_a, sin_a, is_occ=None, quality=1.0,
                             base=(0.95, 1.0, 0.01, 1.0), blur=0.0, max_range=float("inf"), estimator=0,
                             shift_amount=0.0, beam_quality=None):
        """Argument block of slamhip_map_append_scan_q made once for a scan: returns append(pose) -> cell updates (-1
        while updates are deferred)."""
        cfg = ScanAdderCfg(rule, quality, base[0], base[1], base[2], base[3], blur, max_range, estimator, shift_amount)
        rng, cos_a, sin_a = _f64(rng), _f64(cos_a), _f64(sin_a)
        occ = np.ascontiguousarray(is_occ, dtype=np.int32) if is_occ is not None else None
        bq = _f64(beam_quality) if beam_quality is not None else None
        nu = C.c_longlong(0)
        pose3 = (C.c_double * 3)()
        fn = self.L.slamhip_map_append_scan_q
        args = (self.h, map_id, C.pointer(cfg), C.cast(pose3, _dp), rng.size, _d(rng), _d(cos_a), _d(sin_a),
                occ.ctypes.data_as(_ip) if occ is not None else None, _d(bq) if bq is not None else None, C.pointer(nu))
        keep = (cfg, rng, cos_a, sin_a, occ, bq)

        def append(pose, _keep=keep):
            pose3[0], pose3[1], pose3[2] = float(pose[0]), float(pose[1]), float(pose[2])
            rc = fn(*args)
            if rc:
                _check(rc)
            return nu.value
        return append

    def scan_filter_upload(self, map_id, rng, ang, pose, is_occ=None, factor=None, trig_mode=TRIG_RAW, a_min=0.0,
                           a_max=0.0, a_inc=1.0, skip_rate=0, max_range=-1.0, bounded=False, weighting="even"):
        """slamhip_scan_filter_upload: the RAW scan in -- filter_scan, weighting, beam trig and the upload in one call.
        Returns the raw indices of the points kept."""
        rng, ang, pose = _f64(rng), _f64(ang), _f64(pose)
        occ = np.ascontiguousarray(is_occ, dtype=np.int32) if is_occ is not None else None
        fac = _f64(factor) if factor is not None else None
        kept = np.zeros(max(rng.size, 1), np.int32)
        n = C.c_int(0)
        self.L.slamhip_scan_filter_upload.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp, _dp, _ip, _dp, C.c_int, C.c_double,
                                                      C.c_double, C.c_double, _dp, C.c_uint, C.c_double, C.c_int, C.c_int,
                                                      C.POINTER(C.c_int), _ip]
        _check(self.L.slamhip_scan_filter_upload(
            self.h, int(map_id), rng.size, _d(rng), _d(ang), occ.ctypes.data_as(_ip) if occ is not None else None,
            _d(fac) if fac is not None else None, int(trig_mode), a_min, a_max, a_inc, _d(pose), int(skip_rate),
            float(max_range), int(bool(bounded)), {"even": 0, "viny": 1, "ahr": 2}[weighting], C.byref(n),
            kept.ctypes.data_as(_ip)))
        return kept[:n.value].copy()

    def make_raw_scan(self, map_id, rng, ang, is_occ=None, factor=None, trig_mode=TRIG_RAW, a_min=0.0, a_max=0.0,
                      a_inc=1.0, skip_rate=0, max_range=-1.0, bounded=False, weighting="even"):
        """Argument block of slamhip_scan_filter_upload made once for a raw scan (a caller that holds its scans in
        C arrays pays no conversions per call): returns upload(pose) -> number of points kept."""
        rng, ang = _f64(rng), _f64(ang)
        occ = np.ascontiguousarray(is_occ, dtype=np.int32) if is_occ is not None else None
        fac = _f64(factor) if factor is not None else None
        n_kept = C.c_int(0)
        pose3 = (C.c_double * 3)()
        fn = self.L.slamhip_scan_filter_upload
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp, _dp, _ip, _dp, C.c_int, C.c_double, C.c_double, C.c_double,
                       _dp, C.c_uint, C.c_double, C.c_int, C.c_int, C.POINTER(C.c_int), _ip]
        args = (self.h, int(map_id), rng.size, _d(rng), _d(ang), occ.ctypes.data_as(_ip) if occ is not None else None,
                _d(fac) if fac is not None else None, int(trig_mode), a_min, a_max, a_inc,
                C.cast(pose3, _dp), int(skip_rate), float(max_range), int(bool(bounded)),
                {"even": 0, "viny": 1, "ahr": 2}[weighting], C.pointer(n_kept), None)
        keep = (rng, ang, occ, fac)

        def upload(pose, _keep=keep):
            pose3[0], pose3[1], pose3[2] = float(pose[0]), float(pose[1]), float(pose[2])
            rc = fn(*args)
            if rc:
                _check(rc)
            return n_kept.value
        return upload

    def scan_set_angles(self, angle):
        """the angles of the current scan's points: what POSE_TRIG_RAW_EXACT adds the pose heading to"""
        angle = _f64(angle)
        self.L.slamhip_scan_set_angles.argtypes = [C.c_void_p, C.c_int, _dp]
        _check(self.L.slamhip_scan_set_angles(self.h, angle.size, _d(angle)))

    def libm_eval(self, variant, fn, x):
        """glibc's sin (fn 0) / cos (1) / exp (2) as restated in csrc/libm_exact.h, evaluated on the device"""
        x = _f64(x).ravel()
        out = np.zeros(x.size)
        self.L.slamhip_libm_eval.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, _dp, _dp]
        _check(self.L.slamhip_libm_eval(self.h, int(variant), int(fn), x.size, _d(x), _d(out)))
        return out

    def scan_store(self, slot, rng, cos_a, sin_a, weight, factor=None):
        """Keeps a filtered scan resident in HBM (slot 0..4095); scan_select / match jobs refer to it."""
        rng, cos_a, sin_a, weight = _f64(rng), _f64(cos_a), _f64(sin_a), _f64(weight)
        fac = _f64(factor) if factor is not None else np.ones(rng.size)
        _check(self.L.slamhip_scan_store(self.h, int(slot), rng.size, _d(rng), _d(cos_a), _d(sin_a), _d(weight), _d(fac)))

    def scan_select(self, slot):
        rc = self.L.slamhip_scan_select(self.h, int(slot))
        if rc:
            _check(rc)

    # scoring
    def score_poses(self, map_id, cfg, poses):
        poses = _f64(poses).reshape(-1, 3)
        out = np.zeros(poses.shape[0])
        _check(self.L.slamhip_score_poses(self.h, map_id, C.byref(cfg), poses.shape[0], _d(poses),
                                          _d(out)))
        return out

    def score_poses_device(self, map_id, cfg, n, d_poses_ptr, d_scores_ptr):
        _check(self.L.slamhip_score_poses_device(self.h, map_id, C.byref(cfg), n,
                                                 C.c_void_p(d_poses_ptr), C.c_void_p(d_scores_ptr)))

    def shard_init(self, rank, world, unique_id):
        """Joins the RCCL group of `world` contexts (one per GPU)."""
        uid = np.ascontiguousarray(unique_id, dtype=np.uint8)
        assert uid.size == SHARD_ID_BYTES
        _check(self.L.slamhip_shard_init(self.h, rank, world, uid.ctypes.data_as(C.c_void_p)))

    def shard_set_timeout(self, ms):
        """Deadline of every wait on a collective of the RCCL transport (slamhip_shard_set_timeout)."""
        self.L.slamhip_shard_set_timeout.argtypes = [C.c_void_p, C.c_int]
        _check(self.L.slamhip_shard_set_timeout(self.h, int(ms)))

    def shard_attach(self, transport, rank=None, world=None):
        """Joins a group over the caller's own transport (a ctypes slamhip_shard_transport); rank / world default to
        attributes of the same name on the object (tests/loopback.py sets them)."""
        r = rank if rank is not None else getattr(transport, "_rank")
        w = world if world is not None else getattr(transport, "_world")
        _check(self.L.slamhip_shard_attach(self.h, r, w, C.byref(transport)))

    def shard_exchange(self, sends, recvs):
        """slamhip_shard_exchange: sends / recvs are lists of (peer, device pointer, bytes)."""
        def arr(msgs):
            a = (ShardMsg * max(len(msgs), 1))()
            for k, (peer, ptr, nb) in enumerate(msgs):
                a[k].peer, a[k].buf, a[k].bytes = int(peer), int(ptr), int(nb)
            return a
        sa, ra = arr(sends), arr(recvs)
        _check(self.L.slamhip_shard_exchange(self.h, len(sends), sa, len(recvs), ra))

    def shard_p2p_stats(self):
        a, b = C.c_longlong(), C.c_longlong()
        _check(self.L.slamhip_shard_p2p_stats(self.h, C.byref(a), C.byref(b)))
        return dict(exchanges=a.value, bytes_sent=b.value)

    def shard_destroy(self):
        _check(self.L.slamhip_shard_destroy(self.h))

    def shard_info(self):
        r, w = C.c_int(), C.c_int()
        _check(self.L.slamhip_shard_info(self.h, C.byref(r), C.byref(w)))
        return r.value, w.value

    def shard_allgather(self, local, counts):
        """All-gather of per-rank blocks of counts[r] rows (RCCL); returns the rows of all ranks in rank order.
        The row shape is local.shape[1:] -- pass an array shaped (counts[rank], ...) even when counts[rank] is 0, so
        that every rank puts the same element size on the wire."""
        a = np.ascontiguousarray(local)
        cn = np.ascontiguousarray(counts, dtype=np.int32)
        rank, world = self.shard_info()
        assert cn.size == world
        if a.ndim == 0 or a.shape[0] != int(cn[rank]):
            raise ValueError("shard_allgather: local must have shape (counts[rank], ...), got %r for %d rows"
                             % (a.shape, int(cn[rank])))
        row_shape = a.shape[1:]
        row = int(np.prod(row_shape)) if row_shape else 1
        out = np.zeros((int(cn.sum()),) + tuple(row_shape), dtype=a.dtype)
        _check(self.L.slamhip_shard_allgather(self.h, a.ctypes.data_as(C.c_void_p), cn.ctypes.data_as(_ip),
                                              a.dtype.itemsize * row, out.ctypes.data_as(C.c_void_p)))
        return out

    def shard_stats(self):
        a, b = C.c_longlong(), C.c_longlong()
        _check(self.L.slamhip_shard_stats(self.h, C.byref(a), C.byref(b)))
        return dict(collectives=a.value, bytes=b.value)

    def gm_cache_reset(self):
        _check(self.L.slamhip_gm_cache_reset(self.h))

    def gm_cache_get(self):
        xy = np.zeros(2, np.int32)
        p = C.c_double()
        _check(self.L.slamhip_gm_cache_get(self.h, xy.ctypes.data_as(_ip), C.byref(p)))
        return int(xy[0]), int(xy[1]), p.value

    def gm_cache_set(self, cx, cy, prob):
        xy = np.array([cx, cy], np.int32)
        _check(self.L.slamhip_gm_cache_set(self.h, xy.ctypes.data_as(_ip), float(prob)))

    def profile_enable(self, on=True):
        _check(self.L.slamhip_profile_enable(self.h, int(on)))

    def profile_read(self, reset=True):
        ms, la, un = C.c_double(), C.c_longlong(), C.c_longlong()
        _check(self.L.slamhip_profile_read(self.h, C.byref(ms), C.byref(la), C.byref(un), int(reset)))
        return ms.value, la.value, un.value


def _profile_read_map_update(self, reset=True):
    """(ms of the K6 pipelines, calls, (beam, cell) records applied) since the last reset"""
    ms, n, r = C.c_double(), C.c_longlong(), C.c_longlong()
    _check(self.L.slamhip_profile_read_map_update(self.h, C.byref(ms), C.byref(n), C.byref(r), 1 if reset else 0))
    return ms.value, n.value, r.value


Context.profile_read_map_update = _profile_read_map_update


class Matcher:
    """GridScanMatcher counterpart: MC / HC / BF over a Context (slamhip_matcher)."""

    def __init__(self, ctx, kind, cfg, params):
        self.ctx, self.L = ctx, ctx.L
        h = C.c_void_p()
        if kind == "MC":
            _check(self.L.slamhip_matcher_create_mc(ctx.h, C.byref(cfg), int(params[0]), params[1],
                                                    params[2], int(params[3]), int(params[4]),
                                                    C.byref(h)))
        elif kind == "HC":
            _check(self.L.slamhip_matcher_create_hc(ctx.h, C.byref(cfg), int(params[0]), params[1],
                                                    params[2], C.byref(h)))
        elif kind == "BF":
            p = _f64(params)
            _check(self.L.slamhip_matcher_create_bf(ctx.h, C.byref(cfg), _d(p), C.byref(h)))
        else:
            raise ValueError(kind)
        self.h = h
        self._obs = None
        self._obs_cleared = False  # the C side holds no observer of ours (set_observer(NULL) has been called)
        # argument buffers of the per-scan calls, made once (a ctypes object per argument and call was a tenth of a
        # headline match)
        self._ip3, self._d3, self._prob = (C.c_double * 3)(), (C.c_double * 3)(), C.c_double()
        self._st_ll = [C.c_longlong() for _ in range(5)]
        self._st_d = [C.c_double() for _ in range(4)]
        self._st_ll_ref = [C.byref(x) for x in self._st_ll]
        self._st_d_ref = [C.byref(x) for x in self._st_d]
        self._prob_ref = C.byref(self._prob)
        _live["dep"].add(self)
        ctx._deps.add(self)

    def close(self):
        if getattr(self, "h", None):
            self.L.slamhip_matcher_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset_state(self):
        _check(self.L.slamhip_matcher_reset_state(self.h))

    def set_batch(self, n):
        _check(self.L.slamhip_matcher_set_batch(self.h, n))

    def set_device_chain(self, mode, threads=0):
        """Device chains: 2 = hill climbing over the 1-cell OOPE as one co-resident launch (default), 1 = a kernel
        per super-step, 0 = host-driven speculative batches."""
        _check(self.L.slamhip_matcher_set_device_chain(self.h, int(mode), int(threads)))

    def set_tie_check(self, on):
        """Checked default mode (on by default): comparisons the canonical tree sum cannot settle are decided from
        beam-order sums."""
        _check(self.L.slamhip_matcher_set_tie_check(self.h, int(bool(on))))

    def process_scan(self, map_id, init_pose, trace=False):
        """Returns dict(prob, delta[, poses, scores, accepted, n_calls]) -- the trace is what a
        GridScanMatcherObserver sees (on_scan_test / on_pose_update)."""
        ip, d3 = self._ip3, self._d3
        ip[0], ip[1], ip[2] = float(init_pose[0]), float(init_pose[1]), float(init_pose[2])
        rec = None
        if trace:
            rec = dict(poses=[], scores=[], accepted=[])

            def on_test(_u, p, s):
                rec["poses"].append((p[0], p[1], p[2]))
                rec["scores"].append(s)
                rec["accepted"].append(0)

            def on_update(_u, p, s):
                rec["accepted"][-1] = 1

            self._obs = Observer(None, OBS_FN(on_test), OBS_FN(on_update), OBS_FN(0))
            _check(self.L.slamhip_matcher_set_observer(self.h, C.byref(self._obs)))
            self._obs_cleared = False
        elif not self._obs_cleared:
            _check(self.L.slamhip_matcher_set_observer(self.h, None))
            self._obs_cleared = True
        rc = self.L.slamhip_matcher_process_scan(self.h, map_id, ip, d3, self._prob_ref)
        if rc:
            _check(rc)
        out = dict(prob=self._prob.value, delta=np.array((d3[0], d3[1], d3[2])))
        if rec is not None:
            out.update(poses=np.array(rec["poses"]).reshape(-1, 3), scores=np.array(rec["scores"]),
                       accepted=np.array(rec["accepted"], dtype=np.int32),
                       n_calls=len(rec["scores"]))
        return out

    def make_batch(self, jobs):
        """Argument block of process_scan_batch, made once for a list of jobs = dicts(map_id, range, cos_a, sin_a,
        weight[, factor], init_pose): the arrays are kept alive by the returned object."""
        arr = (MatchJob * len(jobs))()
        keep = []
        for k, j in enumerate(jobs):
            for q in range(3):
                arr[k].init_pose[q] = float(j["init_pose"][q])
            arr[k].map_id = int(j["map_id"])
            if j.get("scan_slot") is not None:  # a scan stored in HBM (Context.scan_store)
                arr[k].scan_slot = int(j["scan_slot"])
                continue
            arr[k].scan_slot = -1
            r, c, s_, w = _f64(j["range"]), _f64(j["cos_a"]), _f64(j["sin_a"]), _f64(j["weight"])
            f = _f64(j["factor"]) if j.get("factor") is not None else None
            assert r.size == c.size == s_.size == w.size
            keep += [r, c, s_, w, f]
            arr[k].n = r.size
            arr[k].range, arr[k].cos_a, arr[k].sin_a, arr[k].weight = _d(r), _d(c), _d(s_), _d(w)
            arr[k].factor = _d(f) if f is not None else None
        return dict(arr=arr, keep=keep, n=len(jobs), deltas=np.zeros((len(jobs), 3)), probs=np.zeros(len(jobs)))

    def make_raw_process_scan(self, map_id, rng, ang, is_occ=None, factor=None, trig_mode=TRIG_RAW, a_min=0.0, a_max=0.0,
                              a_inc=1.0, skip_rate=0, max_range=-1.0, bounded=False, weighting="even"):
        """slamhip_matcher_process_raw_scan with its argument block made once for a raw scan (a caller that holds its scans
        in C arrays pays no conversions per call): returns match(pose) -> (points kept, prob); the pose delta of the
        last match is in match.delta (a ctypes array of 3)."""
        rng, ang = _f64(rng), _f64(ang)
        occ = np.ascontiguousarray(is_occ, dtype=np.int32) if is_occ is not None else None
        fac = _f64(factor) if factor is not None else None
        blk = RawScan(rng.size, _d(rng), _d(ang), occ.ctypes.data_as(_ip) if occ is not None else None,
                      _d(fac) if fac is not None else None, int(trig_mode), a_min, a_max, a_inc, int(skip_rate),
                      float(max_range), int(bool(bounded)), {"even": 0, "viny": 1, "ahr": 2}[weighting])
        kept, prob = C.c_int(0), C.c_double(0.0)
        pose3, d3 = (C.c_double * 3)(), (C.c_double * 3)()
        fn = self.L.slamhip_matcher_process_raw_scan
        fn.argtypes = [C.c_void_p, C.c_int, C.POINTER(RawScan), _dp, _dp, _dp, C.POINTER(C.c_int)]
        args = (self.h, int(map_id), C.pointer(blk), C.cast(pose3, _dp), C.cast(d3, _dp), C.cast(C.pointer(prob), _dp),
                C.pointer(kept))
        keep = (rng, ang, occ, fac, blk)

        def match(pose, _keep=keep):
            pose3[0], pose3[1], pose3[2] = float(pose[0]), float(pose[1]), float(pose[2])
            if not self._obs_cleared:
                _check(self.L.slamhip_matcher_set_observer(self.h, None))
                self._obs_cleared = True
            rc = fn(*args)
            if rc:
                _check(rc)
            return kept.value, prob.value
        match.delta = d3
        return match

    def process_scan_batch(self, jobs, trace=False):
        """K independent matches in shared launches (slamhip_matcher_process_scan_batch).  jobs: a list of dicts (see
        make_batch) or a block make_batch returned.  Returns a list of dicts like process_scan's, one per job."""
        blk = jobs if isinstance(jobs, dict) and "arr" in jobs else self.make_batch(jobs)
        recs = None
        if trace:
            recs = [dict(poses=[], scores=[], accepted=[])]

            def on_test(_u, p, s):
                recs[-1]["poses"].append((p[0], p[1], p[2]))
                recs[-1]["scores"].append(s)
                recs[-1]["accepted"].append(0)

            def on_update(_u, p, s):
                recs[-1]["accepted"][-1] = 1

            def on_end(_u, d, s):
                recs.append(dict(poses=[], scores=[], accepted=[]))

            self._obs = Observer(None, OBS_FN(on_test), OBS_FN(on_update), OBS_FN(on_end))
            _check(self.L.slamhip_matcher_set_observer(self.h, C.byref(self._obs)))
            self._obs_cleared = False
        elif not self._obs_cleared:
            _check(self.L.slamhip_matcher_set_observer(self.h, None))
            self._obs_cleared = True
        rc = self.L.slamhip_matcher_process_scan_batch(self.h, blk["n"], blk["arr"], _d(blk["deltas"]), _d(blk["probs"]))
        if rc:
            _check(rc)
        out = []
        for k in range(blk["n"]):
            o = dict(prob=float(blk["probs"][k]), delta=blk["deltas"][k].copy())
            if recs is not None:
                r = recs[k]
                o.update(poses=np.array(r["poses"]).reshape(-1, 3), scores=np.array(r["scores"]),
                         accepted=np.array(r["accepted"], dtype=np.int32), n_calls=len(r["scores"]))
            out.append(o)
        return out

    def batch_stats(self, job):
        a, b, c, d = C.c_longlong(), C.c_longlong(), C.c_longlong(), C.c_int()
        _check(self.L.slamhip_matcher_batch_stats(self.h, job, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return dict(scorer_calls=a.value, poses_evaluated=b.value, super_steps=c.value, on_device_chain=bool(d.value))

    def stats(self):
        (a, b, c, kl, rs), t = self._st_ll, self._st_d
        ra, rt = self._st_ll_ref, self._st_d_ref
        _check(self.L.slamhip_matcher_stats(self.h, ra[0], ra[1], ra[2]))
        _check(self.L.slamhip_matcher_timing(self.h, rt[0], rt[1], rt[2], rt[3]))
        _check(self.L.slamhip_matcher_chain_stats(self.h, ra[3], ra[4]))
        tc = C.c_longlong()
        _check(self.L.slamhip_matcher_tail_stats(self.h, C.byref(tc)))
        return dict(scorer_calls=a.value, poses_evaluated=b.value, launches=c.value,
                    kernels_launched=kl.value, steps_rescored=rs.value, build_us=t[0].value, stage_us=t[1].value, score_us=t[2].value,
                    replay_us=t[3].value, calls_closed_form=tc.value)

    def resident_stats(self):
        """Matches launched in the co-resident form, and how many of them gave up and were redone by the kernel chain."""
        a, b = C.c_longlong(), C.c_longlong()
        _check(self.L.slamhip_matcher_resident_stats(self.h, C.byref(a), C.byref(b)))
        return dict(matches=a.value, gave_up=b.value)


class RawScan(C.Structure):
    """slamhip_raw_scan (include/slamhip.h)"""
    _fields_ = [("n", C.c_int), ("range", _dp), ("angle", _dp), ("is_occ", _ip), ("factor", _dp), ("trig_mode", C.c_int),
                ("a_min", C.c_double), ("a_max", C.c_double), ("a_inc", C.c_double), ("skip_rate", C.c_uint),
                ("max_range", C.c_double), ("bounded", C.c_int), ("weighting", C.c_int)]


class GmappingFilter:
    """GmappingParticleFilter counterpart (slamhip_gmapping): the particles [first, first+count) of
    n_total.  ctx may be None for host-only bookkeeping (weights / resampling of a shard)."""

    def __init__(self, ctx, params, n_total, seeds, first=0, count=None):
        self.L = ctx.L if ctx is not None else load()
        self.ctx = ctx
        self.n_total, self.first = n_total, first
        self.count = n_total - first if count is None else count
        sd = np.ascontiguousarray(seeds, dtype=np.uint32)
        assert sd.size == self.count, "one seed per local particle"
        h = C.c_void_p()
        _check(self.L.slamhip_gmapping_create(ctx.h if ctx is not None else None, C.byref(params),
                                              n_total, first, self.count,
                                              sd.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(h)))
        self.h = h
        _live["dep"].add(self)
        if ctx is not None:
            ctx._deps.add(self)

    def close(self):
        if getattr(self, "h", None):
            self.L.slamhip_gmapping_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def predict_match(self, map_id, rng, ang, is_occ, odom_delta):
        rng, ang, d = _f64(rng), _f64(ang), _f64(odom_delta)
        occ = np.ascontiguousarray(is_occ if is_occ is not None else np.ones(rng.size), dtype=np.int32)
        raw = np.zeros(self.count)
        _check(self.L.slamhip_gmapping_predict_match(self.h, map_id, rng.size, _d(rng), _d(ang),
                                                     occ.ctypes.data_as(_ip), _d(d), _d(raw)))
        return raw

    def plan_resample(self, all_raw_weights, seed):
        w = _f64(all_raw_weights)
        assert w.size == self.n_total
        req = C.c_int(0)
        idx = np.zeros(self.n_total, np.uint32)
        _check(self.L.slamhip_gmapping_plan_resample(self.h, _d(w), seed, C.byref(req),
                                                     idx.ctypes.data_as(C.POINTER(C.c_uint))))
        return bool(req.value), idx

    def blob_size(self):
        return int(self.L.slamhip_gmapping_blob_size())

    def export(self):
        buf = np.zeros(self.count * self.blob_size(), np.uint8)
        _check(self.L.slamhip_gmapping_export(self.h, buf.ctypes.data_as(C.c_void_p)))
        return buf

    def import_(self, all_blobs, idx):
        b = np.ascontiguousarray(all_blobs, dtype=np.uint8)
        assert b.size == self.n_total * self.blob_size()
        ix = np.ascontiguousarray(idx, dtype=np.uint32)
        _check(self.L.slamhip_gmapping_import(self.h, b.ctypes.data_as(C.c_void_p),
                                              ix.ctypes.data_as(C.POINTER(C.c_uint))))

    def step(self, map_id, rng, ang, is_occ, odom_delta, resample_seed):
        rng, ang, d = _f64(rng), _f64(ang), _f64(odom_delta)
        occ = np.ascontiguousarray(is_occ, dtype=np.int32) if is_occ is not None else None  # (null: every point occupied)
        res = C.c_int(0)
        idx = np.zeros(self.n_total, np.uint32)
        _check(self.L.slamhip_gmapping_step(self.h, map_id, rng.size, _d(rng), _d(ang),
                                            occ.ctypes.data_as(_ip) if occ is not None else None, _d(d), resample_seed,
                                            C.byref(res), idx.ctypes.data_as(C.POINTER(C.c_uint))))
        return bool(res.value), idx

    def step_sharded(self, map_id, rng, ang, is_occ, odom_delta, resample_seed):
        """One scan on this shard, RCCL collectives included (Context.shard_init first)."""
        rng, ang, d = _f64(rng), _f64(ang), _f64(odom_delta)
        occ = np.ascontiguousarray(is_occ, dtype=np.int32) if is_occ is not None else None
        res = C.c_int(0)
        idx = np.zeros(self.n_total, np.uint32)
        _check(self.L.slamhip_gmapping_step_sharded(self.h, map_id, rng.size, _d(rng), _d(ang),
                                                    occ.ctypes.data_as(_ip) if occ is not None else None, _d(d), resample_seed,
                                                    C.byref(res), idx.ctypes.data_as(C.POINTER(C.c_uint))))
        return bool(res.value), idx

    # a step in phases (sharded filters with a caller-side collective)
    def set_shard_chain(self, on=True):
        _check(self.L.slamhip_gmapping_set_shard_chain(self.h, 1 if on else 0))

    def match_begin(self, map_id, rng, ang, is_occ, odom_delta):
        rng, ang, d = _f64(rng), _f64(ang), _f64(odom_delta)
        occ = np.ascontiguousarray(is_occ, dtype=np.int32) if is_occ is not None else None
        _check(self.L.slamhip_gmapping_match_begin(self.h, map_id, rng.size, _d(rng), _d(ang),
                                                   occ.ctypes.data_as(_ip) if occ is not None else None, _d(d)))

    def carry_record(self):
        r = CarryRecord()
        _check(self.L.slamhip_gmapping_carry_record(self.h, C.byref(r)))
        return r

    def carry_fix(self, records, rank):
        arr = (CarryRecord * len(records))(*records)
        ch = C.c_int(0)
        _check(self.L.slamhip_gmapping_carry_fix(self.h, arr, len(records), rank, C.byref(ch)))
        return bool(ch.value)

    def carry_commit(self, records):
        arr = (CarryRecord * len(records))(*records)
        _check(self.L.slamhip_gmapping_carry_commit(self.h, arr, len(records)))

    def match_finish(self):
        raw = np.zeros(self.count)
        _check(self.L.slamhip_gmapping_match_finish(self.h, _d(raw)))
        return raw

    def set_map_update(self, enable=True, base=(0.95, 1.0, 0.01, 1.0), blur=0.0, max_range=float("inf"),
                       estimator=0, shift_amount=0.0):
        """The reference's full step: each matching particle appends its scan to the shared map."""
        if not enable:
            _check(self.L.slamhip_gmapping_set_map_update(self.h, None))
            return
        cfg = ScanAdderCfg(RULE_GMAPPING, 1.0, base[0], base[1], base[2], base[3], blur, max_range,
                           estimator, shift_amount)
        _check(self.L.slamhip_gmapping_set_map_update(self.h, C.byref(cfg)))

    def enable_particle_maps(self, map_id, extent_tiles, pool_tiles, base=(0.95, 1.0, 0.01, 1.0), blur=0.0,
                             max_range=float("inf"), estimator=0, shift_amount=0.0):
        """Per-particle copy-on-write maps (tile pool) seeded from the bound dense window `map_id`."""
        cfg = ScanAdderCfg(RULE_GMAPPING, 1.0, base[0], base[1], base[2], base[3], blur, max_range,
                           estimator, shift_amount)
        _check(self.L.slamhip_gmapping_enable_particle_maps(self.h, map_id, C.byref(cfg), extent_tiles,
                                                            pool_tiles))

    def particle_maps_append(self, particles, poses, rng, ang, is_occ=None):
        """Appends the raw scan to the maps of the listed local particles, each from its own pose (one batched
        K6); returns the number of cell updates."""
        pt = np.ascontiguousarray(particles, dtype=np.int32)
        ps, rng, ang = _f64(poses), _f64(rng), _f64(ang)
        assert ps.size == 3 * pt.size
        occ = np.ascontiguousarray(is_occ, dtype=np.int32) if is_occ is not None else None
        nu = C.c_longlong(0)
        _check(self.L.slamhip_gmapping_particle_maps_append(self.h, pt.size, pt.ctypes.data_as(_ip), _d(ps), rng.size,
                                                            _d(rng), _d(ang),
                                                            occ.ctypes.data_as(_ip) if occ is not None else None,
                                                            C.byref(nu)))
        return nu.value

    def particle_map(self, particle, x0, y0, w, h):
        """(payload[h, w, 3], counters[h, w, 2]) of the external window of one particle's map."""
        pay, aux = np.zeros((h, w, 3)), np.zeros((h, w, 2))
        _check(self.L.slamhip_gmapping_particle_map_download(self.h, particle, x0, y0, w, h, _d(pay), _d(aux)))
        return pay, aux

    def export_particle_map(self, particle):
        """The LOCAL particle's map as one uint8 array (tile positions + tiles) for another rank."""
        sz = C.c_size_t(0)
        _check(self.L.slamhip_gmapping_particle_map_export_size(self.h, particle, C.byref(sz)))
        buf = np.zeros(sz.value, np.uint8)
        _check(self.L.slamhip_gmapping_particle_map_export(self.h, particle, buf.ctypes.data_as(C.c_void_p),
                                                           buf.size))
        return buf

    def import_maps(self, all_blobs, idx, remote):
        """Resampling with per-particle maps: `remote` = {global source particle: exported uint8 array}
        for every source that lives on another rank."""
        b = np.ascontiguousarray(all_blobs, dtype=np.uint8)
        ix = np.ascontiguousarray(idx, dtype=np.uint32)
        keys = sorted(remote)
        src = np.ascontiguousarray(keys, dtype=np.int32)
        bufs = [np.ascontiguousarray(remote[k], dtype=np.uint8) for k in keys]
        ptrs = (C.c_void_p * max(len(bufs), 1))(*[bb.ctypes.data for bb in bufs])
        _check(self.L.slamhip_gmapping_import_maps(self.h, b.ctypes.data_as(C.c_void_p),
                                                   ix.ctypes.data_as(C.POINTER(C.c_uint)), len(keys),
                                                   src.ctypes.data_as(_ip), ptrs))

    def particle_map_stats(self):
        v = [C.c_longlong() for _ in range(5)]
        _check(self.L.slamhip_gmapping_particle_map_stats(self.h, *[C.byref(x) for x in v]))
        return dict(tiles_in_use=v[0].value, tiles_shared=v[1].value, bytes=v[2].value, cow_copies=v[3].value,
                    cell_updates=v[4].value)

    def set(self, poses=None, weights=None):
        p = _f64(poses) if poses is not None else None
        w = _f64(weights) if weights is not None else None
        _check(self.L.slamhip_gmapping_set(self.h, _d(p) if p is not None else None,
                                           _d(w) if w is not None else None))

    def state(self):
        poses, w, ms = np.zeros((self.count, 3)), np.zeros(self.count), np.zeros(self.count, np.int32)
        _check(self.L.slamhip_gmapping_get(self.h, _d(poses), _d(w), ms.ctypes.data_as(_ip)))
        return poses, w, ms

    def match_abort(self):
        _check(self.L.slamhip_gmapping_match_abort(self.h))

    def migration_stats(self):
        a, b = C.c_longlong(), C.c_longlong()
        _check(self.L.slamhip_gmapping_migration_stats(self.h, C.byref(a), C.byref(b)))
        return dict(maps_received=a.value, tile_bytes_sent=b.value)

    def stats(self):
        v = [C.c_longlong() for _ in range(4)]
        _check(self.L.slamhip_gmapping_stats(self.h, *[C.byref(x) for x in v]))
        return dict(scorer_calls=v[0].value, poses_evaluated=v[1].value, launches=v[2].value,
                    carry_reruns=v[3].value)
