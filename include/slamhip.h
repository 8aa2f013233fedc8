/* include/slamhip.h -- C-ABI of the MI355X (gfx950) scan-matching / particle-likelihood engine.
 *
 * Drop-in boundary for ONE hot path of OSLL/slam-constructor: the GridScanMatcher (MC/HC/BF)
 * accept loop and the ScanProbabilityEstimator scoring it calls for every candidate pose and
 * every GMapping particle, plus the particle weight/normalise/resample step.
 *
 * The reference has no FFI: the path sits behind C++ virtual interfaces in headers.  Each entry
 * point below names the reference interface (file:line, relative to the reference root) whose
 * work it replaces; INTEGRATION.md shows the adapter classes (deriving from the reference's
 * GridScanMatcher / ScanProbabilityEstimator) that bind them.
 *
 * Conventions: plain C, pointers + sizes, no C++/torch types.  Every function returns 0 on
 * success or a negative slamhip_status; slamhip_last_error() gives the text.  One context =
 * one GPU + one HIP stream + one caller thread (the reference is single-threaded:
 * src/ros/single_hypoth_slam_node.cpp:63); contexts are independent (one per GPU when
 * particles are sharded).  There is NO CPU fallback: without a usable GPU the calls fail.
 *
 * All arithmetic is IEEE double in the reference's operation order (built with
 * -ffp-contract=off); see DESIGN.md for the exact-parity contract.
 */
#ifndef SLAMHIP_H
#define SLAMHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLAMHIP_VERSION 100

typedef enum {
  SLAMHIP_OK = 0,
  SLAMHIP_ERR_INVALID = -1,   /* bad argument / unknown id */
  SLAMHIP_ERR_HIP = -2,       /* HIP runtime error (text in slamhip_last_error) */
  SLAMHIP_ERR_NO_DEVICE = -3, /* no usable GPU: the product path never falls back to the CPU */
  SLAMHIP_ERR_STATE = -4,     /* call order (e.g. scoring before a scan/map was uploaded) */
  SLAMHIP_ERR_UNSUPPORTED = -5,
  SLAMHIP_ERR_TIMEOUT = -6    /* a collective of the shard group did not complete within the group's deadline */
} slamhip_status;

/* Cell payload models mirrored in HBM:
 *  OCC      1 x f64  prob_occ            GridCell / MeanProbabilityCell / AffineQualityMergeCell
 *                                        (src/core/maps/grid_cell.h:33-35, naive_grid_cells.h:6-44)
 *  TBM      4 x f64  (u, e, o, c)        TbmBaseCell belief (src/core/maps/tbm_grid_cells.h:21-35)
 *  GMAPPING 3 x f64  (prob_occ, obst.x, obst.y)  GmappingBaseCell
 *                                        (src/slams/gmapping/gmapping_grid_cell.h:9-43) */
enum { SLAMHIP_CELL_OCC = 0, SLAMHIP_CELL_TBM = 1, SLAMHIP_CELL_GMAPPING = 2 };
/* OccupancyObservationProbabilityEstimator kinds
 * (src/core/scan_matchers/occupancy_observation_probability.h:12-99,
 *  src/slams/gmapping/gmapping_occupancy_observation_pe.h:11-45) */
enum { SLAMHIP_OOPE_OBSTACLE = 0, SLAMHIP_OOPE_MAX = 1, SLAMHIP_OOPE_MEAN = 2,
       SLAMHIP_OOPE_OVERLAP = 3, SLAMHIP_OOPE_GMAPPING = 4 };
/* ObservationImpactEstimator kinds (src/core/scan_matchers/observation_impact_estimators.h:14-28) */
enum { SLAMHIP_OIE_DISCREPANCY = 0, SLAMHIP_OIE_OCCUPANCY = 1 };
/* Per-pose summation order of sum_i p_i*w_i*factor_i (weighted_mean_point_probability_spe.h:124):
 *  TREE256    canonical, launch-shape independent tree (default; ~1e-16 from the reference)
 *  SEQUENTIAL the reference's beam-order sum, bit-exact (slower: second pass) */
enum { SLAMHIP_SUM_TREE256 = 0, SLAMHIP_SUM_SEQUENTIAL = 1 };
/* Where sin/cos of the pose heading come from (TrigonometryProvider::set_base_angle,
 * src/core/trigonometry_utils.h:31-33,57-60): host libm (bit-exact with the reference) or the
 * device's sincos (default for throughput; differs from glibc in the last ulp at most). */
enum { SLAMHIP_POSE_TRIG_DEVICE = 0, SLAMHIP_POSE_TRIG_HOST = 1,
       /* RAW_EXACT: the reference's DEFAULT provider bit for bit -- RawTrigonometryProvider evaluates std::cos / std::sin
        * (theta + a) per beam and pose (trigonometry_utils.h:17-35; use_trig_cache = false, src/ros/init_utils.h:56-58).
        * The device evaluates glibc 2.35's sin / cos restated operation for operation (csrc/libm_exact.h: the build of
        * them this host's libm runs, slamhip_libm_variant) on theta + a; needs the beam angles (slamhip_scan_set_angles
        * or slamhip_scan_filter_upload).  Host-driven and slow: the mode results are checked against, not a fast path. */
       SLAMHIP_POSE_TRIG_RAW_EXACT = 2 };
enum { SLAMHIP_TRIG_RAW = 0, SLAMHIP_TRIG_CACHED = 1 };

typedef struct slamhip_ctx slamhip_ctx;
typedef struct slamhip_matcher slamhip_matcher;

/* ScanProbabilityEstimator configuration = what init_spe/init_oope/init_oie build
 * (src/utils/init_scan_matching.h:27-113) + ScanProbabilityEstimator::SPEParams
 * (src/core/scan_matchers/grid_scan_matcher.h:87-91). */
typedef struct {
  int oope;              /* SLAMHIP_OOPE_* */
  int oie;               /* SLAMHIP_OIE_* */
  double area[4];        /* sp_analysis_area: bot, top, left, right (all 0 = point) */
  double gm_fullness_th; /* GmappingOccupancyObservationPE(fullness_th, window) */
  int gm_window;
  int sum_order;         /* SLAMHIP_SUM_* */
  int pose_trig;         /* SLAMHIP_POSE_TRIG_* */
} slamhip_spe_cfg;

/* ---------------------------------------------------------------- context */
const char *slamhip_last_error(void);
int slamhip_device_count(int *count);
int slamhip_ctx_create(int device, slamhip_ctx **out);
int slamhip_ctx_destroy(slamhip_ctx *ctx);
int slamhip_ctx_synchronize(slamhip_ctx *ctx);
/* hipStream_t of the context, for callers that enqueue their own work around ours */
void *slamhip_ctx_stream(slamhip_ctx *ctx);
/* Options of a context.  Every option switches between execution paths that give the SAME results (the parity tests
 * and the measurements in DESIGN.md run both sides); the library reads no environment variables.  Set them before the
 * objects they concern are created or used. */
enum {
  SLAMHIP_OPT_LOW_LATENCY = 0,   /* 1 (default): results come back through pinned completion words the host polls,
                                  * poses are read over PCIe by the kernels; 0: stream synchronisation */
  SLAMHIP_OPT_STAGE_POSES = 1,   /* 1: pose batches are copied to HBM before a scoring kernel reads them; 0 (default) */
  SLAMHIP_OPT_FILTER_CHAINS = 2, /* 1 (default): a filter step runs one accept chain per particle on the device;
                                  * 0: the host-driven lock-step jobs */
  SLAMHIP_OPT_K6_PATH = 3,       /* single-scan map update: 0 (default) the fastest pipeline that applies (gather,
                                  * else counting sort, else radix sort), 1 counting sort, 2 radix sort */
  SLAMHIP_OPT_K6_BATCH_FAST = 4, /* batched map update: 1 (default) free observations of zero-mean cells are settled
                                  * with atomics and only the rest is sorted; 0: every record is sorted into its chain */
  SLAMHIP_OPT_RESIDENT_CHAINS = 6, /* 1 (default): the filter's per-particle accept chains run as ONE launch of
                                    * co-resident workgroups when they all fit the device (csrc/hc_resident_gm.hip);
                                    * 0: a kernel per super-step (csrc/hc_chain.hip).  Same results either way. */
  SLAMHIP_OPT_K6_BATCH_KEY64 = 5, /* batched map update: 1 forces the 8-byte (particle, cell) keys of very large
                                   * batches; 0 (default): by size */
  SLAMHIP_OPT_TBM_PLANE = 7,     /* 1 (default): the 1-cell scorers over a TBM map gather the cell's per-beam
                                  * probability -- a pure function of the cell (tbm_grid_cells.h:21-35 with the fixed
                                  * observation of the scorer) -- from an 8-byte plane every writer of the map keeps,
                                  * instead of the 32-byte cell and its belief arithmetic per (pose, beam); 0: from the
                                  * cell.  The same operations either way: the same bits. */
  SLAMHIP_OPT_INERT_TAIL = 8     /* the tail of a hill-climbing chain on the device (1-cell OOPE).  The reference's enumerator
                                  * stops at a count of failed rounds, not at convergence
                                  * (hill_climbing_scan_matcher.h:83-101), the steps halved at every failure.
                                  * 1: once the steps are below half an ulp of every pose coordinate, every candidate
                                  * of every further round IS the best pose, bit for bit: the same score, `best < score`
                                  * false (pose_enumeration_scan_matcher.h:58) -- the chain ends there; the remaining
                                  * 6 x (limit - failed rounds) + 1 scorer calls are reported to the observer and
                                  * counted, not scored again.  2 (default): also once a tree's walk has accepted
                                  * nothing and the steps have become so small that every beam of every further
                                  * candidate provably ends in the same cell as under the best pose (a per-beam bound
                                  * on the end point's movement against its distance from the cell's edges, made by the
                                  * chain's idle bookkeeping workgroup: csrc/hc_resident.hip "certificate"): the same
                                  * cells, the same terms, the same score (co-resident form only; the chain of kernels
                                  * stays at 1).
                                  * 0: every call scored.  Same traces, results and counts in all three. */
};
int slamhip_ctx_set_option(slamhip_ctx *ctx, int option, int value);
int slamhip_ctx_get_option(slamhip_ctx *ctx, int option, int *value);

/* ---------------------------------------------------------------- map mirror
 * Replaces GridMap::operator[] on the scoring path (src/core/maps/grid_map.h:62,
 * plain_grid_map.h:27-42,69-73, lazy_tiled_grid_map.h:47-55,140-147): the adapter mirrors the
 * cells into a dense pitched HBM window.  Coordinates are INTERNAL (external + origin,
 * regular_squares_grid.h:141-143); reads outside the window return `unknown_payload`, exactly
 * like Unbounded*GridMap::operator[] returns its prototype cell. */
int slamhip_map_bind(slamhip_ctx *ctx, int map_id, int cell_model, int width, int height,
                     int origin_x, int origin_y, double scale, const double *unknown_payload);
/* payload: host, row-major [h][w][stride]; window at internal (x0, y0) */
int slamhip_map_upload_window(slamhip_ctx *ctx, int map_id, int x0, int y0, int w, int h,
                              const double *payload);
/* GridMap::update / reset (src/core/maps/grid_map.h:41-60) forwarded as a dirty-cell log:
 * n cells, coords_xy internal (2 ints each), payloads n*stride doubles */
int slamhip_map_apply_dirty(slamhip_ctx *ctx, int map_id, int n, const int *coords_xy,
                            const double *payloads);
/* Unbounded maps move their origin when they grow (plain_grid_map.h:133-173): re-bind keeps
 * the old cells at their shifted place. */
int slamhip_map_release(slamhip_ctx *ctx, int map_id);
/* An UNBOUNDED map in HBM (UnboundedPlainGridMap, src/core/maps/plain_grid_map.h:52-176): with on = 1 a
 * slamhip_map_append_scan that reaches beyond the window grows it first -- a re-bind that keeps every cell at its
 * external coordinates, the new area holding the unknown payload -- instead of failing.  The reference grows
 * inside GridMap::update, cell by cell, by ensure_inside's Expansion_Rate rule (:133-173); the window here is a
 * superset of that one with the same cells.  This is what lets a single-hypothesis world keep its map on the GPU
 * (host/slamhip_resident_world.h). */
int slamhip_map_set_auto_grow(slamhip_ctx *ctx, int map_id, int on);
/* geometry of a bound window (RegularSquaresGrid::width/height/origin/scale, regular_squares_grid.h:26-38,141-143)
 * and how often it has grown; any out pointer may be NULL */
int slamhip_map_info(slamhip_ctx *ctx, int map_id, int *cell_model, int *width, int *height, int *origin_x,
                     int *origin_y, double *scale, long long *times_grown);
/* read back a window (tests / debugging) */
int slamhip_map_download_window(slamhip_ctx *ctx, int map_id, int x0, int y0, int w, int h,
                                double *payload_out);

/* ---------------------------------------------------------------- map update (kernel K6)
 * Replaces GridMapScanAdder::append_scan (src/core/maps/grid_map_scan_adders.h:54-75) with
 * WallDistanceBlurringScanAdder::handle_scan_point (:138-172) and ConstOccupancyEstimator
 * (const_occupancy_estimator.h:6-17) on the HBM mirror: per beam the 4-connected ray walk
 * (regular_squares_grid.h:56-101), per cell the `cell += observation` of the map's cell class, in
 * the reference's update order (records sorted by cell, stable in beam order).
 * rule = which GridCell subclass the OCC/TBM/GMAPPING payload belongs to. */
enum { SLAMHIP_RULE_LAST = 0,     /* GridCell::operator+= (grid_cell.h:27-30): last write wins */
       SLAMHIP_RULE_AFFINE = 1,   /* AffineQualityMergeCell (naive_grid_cells.h:14-20) */
       SLAMHIP_RULE_MEAN = 2,     /* MeanProbabilityCell (naive_grid_cells.h:33-40) */
       SLAMHIP_RULE_TBM = 3,      /* TbmBaseCell (tbm_grid_cells.h:12-19) */
       SLAMHIP_RULE_GMAPPING = 4  /* GmappingBaseCell (gmapping_grid_cell.h:20-33) */ };
typedef struct {
  int rule;
  double scan_quality;                           /* append_scan's scan_quality (x IdleOMQE = 1) */
  double base_occupied_prob, base_occupied_qual; /* slam/occupancy_estimator/base_occupied/{prob,qual} */
  double base_empty_prob, base_empty_qual;       /* .../base_empty/{prob,qual} (init_occupancy_mapping.h:48-51) */
  double blur;                                   /* slam/mapping/blur, metres; < 0 = dynamic */
  double max_range;                              /* slam/mapping/max_range (infinity = unlimited) */
  int occupancy_estimator;  /* slam/occupancy_estimator/type: 0 const (const_occupancy_estimator.h:6-17),
                             * 1 area (AreaOccupancyEstimator, area_occupancy_estimator.h:27-240) */
  double area_shift_amount; /* Q27: the area estimator freezes low_qual * side of the FIRST cell it
                             * ever sees in a function-local static (area_occupancy_estimator.h:68-72);
                             * 0 = 0.01 * scale */
} slamhip_scan_adder_cfg;
/* RAW scan points (range, cos/sin of their angle as for slamhip_scan_upload, is_occupied flag; may
 * be NULL = all occupied).  Every touched cell must lie inside the bound window (grow it with
 * slamhip_map_bind first, or let it grow: slamhip_map_set_auto_grow), otherwise SLAMHIP_ERR_STATE.
 * n_updates = number of cell updates. */
int slamhip_map_append_scan(slamhip_ctx *ctx, int map_id, const slamhip_scan_adder_cfg *cfg,
                            const double pose[3], int n, const double *range, const double *cos_a,
                            const double *sin_a, const int *is_occ, long long *n_updates);
/* The same with a PER-POINT observation quality: the update quality of a cell a beam touches is scan_quality x
 * quality[i] -- GridMapScanAdder::append_scan multiplies by ObservationMappingQualityEstimator::quality(points, pt_i)
 * (grid_map_scan_adders.h:17-43,66-71).  quality = NULL is IdleOMQE (1.0); slamhip_omqe_quality gives the values of
 * the estimators init_omqe builds (init_occupancy_mapping.h:64-80: "idle", "ahr").  The AffineQualityMergeCell,
 * MeanProbabilityCell and TbmBaseCell rules read it; GridCell and GmappingBaseCell ignore an observation's quality. */
int slamhip_map_append_scan_q(slamhip_ctx *ctx, int map_id, const slamhip_scan_adder_cfg *cfg,
                              const double pose[3], int n, const double *range, const double *cos_a,
                              const double *sin_a, const int *is_occ, const double *quality, long long *n_updates);
/* The same from the points' ANGLES, with the reference's DEFAULT trig provider bit for bit: RawTrigonometryProvider
 * evaluates std::cos / std::sin(pose heading + angle) per point (trigonometry_utils.h:17-35; append_scan sets the base
 * angle to the pose's, grid_map_scan_adders.h:61).  The host's libm makes the 2 n values once per scan; the other two
 * entry points take the provider's own table (cos a, sin a) and add the heading by the cached provider's formulas. */
int slamhip_map_append_scan_raw(slamhip_ctx *ctx, int map_id, const slamhip_scan_adder_cfg *cfg, const double pose[3],
                                int n, const double *range, const double *angle, const int *is_occ,
                                const double *quality, long long *n_updates);
/* host helper: ObservationMappingQualityEstimator::quality of every point of a scan (n values): kind 0 IdleOMQE,
 * 1 AngleHistogramResiprocalOMQE = 1 / AngleHistogram::value (grid_map_scan_adders.h:24-43,
 * src/core/features/angle_histogram.h:17-96), over the scan append_scan is given (all its points) */
int slamhip_omqe_quality(int kind, int n, const double *range, const double *angle, double *out);
/* Map updates queued, not awaited (default off).  While on, slamhip_map_append_scan on the zero-copy path returns as
 * soon as the update's kernels are queued on the context's stream -- the scan arrays are consumed before it returns,
 * *n_updates is -1 -- and everything else the context does (matches, scores, downloads, further updates) is ordered
 * behind them: the pose of a scan can be handed on while the GPU still writes that scan into the map
 * (SingleStateHypothesisLaserScanGridWorld::handle_observation, single_state_hypothesis_laser_scan_grid_world.h:52-65,
 * returns only after append_scan).  slamhip_map_drain waits for the queued updates and reports the sum of their cell
 * updates; a failure on the device (a beam outside a window that may not grow) surfaces there as SLAMHIP_ERR_STATE.
 * At most 64 updates stay queued (the 65th drains first); set_deferred(0) drains; a GMapping filter step on the
 * same context drains what is queued into its own count. */
int slamhip_map_set_deferred(slamhip_ctx *ctx, int on);
int slamhip_map_drain(slamhip_ctx *ctx, long long *n_updates);
/* update counters of a window (MEAN: n; GMAPPING: hits, tries) -- tests / debugging */
int slamhip_map_download_aux(slamhip_ctx *ctx, int map_id, int x0, int y0, int w, int h, double *out);

/* ---------------------------------------------------------------- scan
 * The FILTERED scan the scorer iterates (LaserScan2D after
 * WeightedMeanPointProbabilitySPE::filter_scan, weighted_mean_point_probability_spe.h:75-95),
 * flattened: per point range, cos/sin of its own angle as the scan's TrigonometryProvider
 * tabulates them, weight (ScanPointWeighting::weight, :21-60) and factor (sensor_data.h:74-75).
 * The arrays are consumed before the call returns; the copy to HBM is queued on the context's stream (no wait for
 * work that is still running there, e.g. a queued map update), ahead of every later score or match. */
int slamhip_scan_upload(slamhip_ctx *ctx, int n, const double *range, const double *cos_a,
                        const double *sin_a, const double *weight, const double *factor);
/* Scans kept RESIDENT in HBM: store copies a filtered scan (arrays as above) into slot `slot` (0..4095) of the
 * context, select makes a stored scan the one the following scores and matches read -- a pointer swap on the
 * host, nothing moves.  For callers that match the same scans repeatedly or several at once
 * (slamhip_matcher_process_scan_batch takes slots as well); a later slamhip_scan_upload replaces the selection,
 * not the stored scans. */
int slamhip_scan_store(slamhip_ctx *ctx, int slot, int n, const double *range, const double *cos_a,
                       const double *sin_a, const double *weight, const double *factor);
int slamhip_scan_select(slamhip_ctx *ctx, int slot);
/* The angles of the CURRENT scan's points (ScanPoint2D::angle, sensor_data.h:60-70), n = its point count: what
 * SLAMHIP_POSE_TRIG_RAW_EXACT adds the pose heading to.  Dropped by the next upload / select; slamhip_scan_filter_upload
 * sets them itself. */
int slamhip_scan_set_angles(slamhip_ctx *ctx, int n, const double *angle);
/* Which build of glibc's sin / cos / exp this host's libm runs, found by calling it on arguments where the builds
 * differ: 1 = the FMA build (x86-64 with AVX2 + FMA usable), 0 = the plain build, -1 = neither (another libm): the
 * exact modes (RAW_EXACT, the GMapping OOPE under SLAMHIP_SUM_SEQUENTIAL) then fail with SLAMHIP_ERR_UNSUPPORTED. */
int slamhip_libm_variant(int *variant);
/* the restated functions evaluated ON THE DEVICE (fn 0 sin, 1 cos, 2 exp; variant as above), host arrays in and out:
 * lets a caller (and the tests) confirm device == host libm on its own arguments */
int slamhip_libm_eval(slamhip_ctx *ctx, int variant, int fn, int n, const double *x, double *out);
/* host helpers building cos_a/sin_a: RawTrigonometryProvider (trigonometry_utils.h:17-35) ... */
int slamhip_beam_trig_raw(int n, const double *angle, double *cos_out, double *sin_out);
/* ... and CachedTrigonometryProvider::update + index lookup (trigonometry_utils.h:45-78) */
int slamhip_beam_trig_cached(int n, const double *angle, double a_min, double a_max, double a_inc,
                             double *cos_out, double *sin_out);
/* WeightedMeanPointProbabilitySPE::filter_scan on the host (once per scan, :75-95,136-141).
 * bounded: 1 for PlainGridMap/LazyTiledGridMap (has_cell tests the window), 0 for Unbounded*.
 * cos_a/sin_a as above for the RAW scan; kept_idx receives the raw indices kept. */
int slamhip_filter_scan(int n, const double *range, const double *angle, const int *is_occ,
                        int trig_mode, double a_min, double a_delta, int table_n,
                        const double *tab_sin, const double *tab_cos, const double pose[3],
                        unsigned skip_rate, double max_range, int bounded, int width, int height,
                        int origin_x, int origin_y, double scale, int *kept_idx, int *kept_n);
/* filter_scan + weights + beam trig + slamhip_scan_upload in one call, from the RAW scan (range, angle, is_occ or
 * NULL = all occupied, factor or NULL = all 1.0): what a GridScanMatcher does with a scan before the first candidate
 * is scored (weighted_mean_point_probability_spe.h:75-95; ScanPointWeighting :21-60; trig providers
 * trigonometry_utils.h:17-84).  `bounded`: the map map_id is a PlainGridMap / LazyTiledGridMap whose has_cell()
 * tests the window; 0 for the Unbounded* maps, for which no end point is computed at all (map_id is then not
 * looked at).  Quantities that depend on the beam ANGLES only are kept inside the context until the angle array
 * changes.  weighting: 0 even, 1 viny, 2 ahr; kept_n (may be NULL) = points kept, kept_idx (may be NULL, room for n)
 * their raw indices.  With no point kept nothing is uploaded and scoring fails with SLAMHIP_ERR_STATE (the
 * reference scores such a scan NaN). */
int slamhip_scan_filter_upload(slamhip_ctx *ctx, int map_id, int n, const double *range, const double *angle,
                               const int *is_occ, const double *factor, int trig_mode, double a_min, double a_max,
                               double a_inc, const double pose[3], unsigned skip_rate, double max_range, int bounded,
                               int weighting, int *kept_n, int *kept_idx);
/* ScanPointWeighting::weight for a filtered scan: kind 0 even, 1 viny, 2 ahr */
int slamhip_scan_weights(int kind, int n, const double *range, const double *angle, double *out);

/* ---------------------------------------------------------------- scoring (kernels K1/K2/K3)
 * Replaces ScanProbabilityEstimator::estimate_scan_probability
 * (src/core/scan_matchers/grid_scan_matcher.h:128-131; WMPP implementation
 * weighted_mean_point_probability_spe.h:97-133) for a BATCH of poses: poses_xyt = n_poses x
 * (x, y, theta) host doubles, scores_out n_poses host doubles (NaN when sum of weights is 0). */
int slamhip_score_poses(slamhip_ctx *ctx, int map_id, const slamhip_spe_cfg *cfg, int n_poses,
                        const double *poses_xyt, double *scores_out);
/* Same with device-resident poses/scores, asynchronous on the context stream (sweep mode) */
int slamhip_score_poses_device(slamhip_ctx *ctx, int map_id, const slamhip_spe_cfg *cfg,
                               int n_poses, const double *d_poses_xyt, double *d_scores_out);
/* GMapping OOPE run-cache state carried between calls (gmapping_occupancy_observation_pe.h:21-24,
 * 36-37,43-44): {cell x, cell y} + cached probability (-1 = empty).  Used by score_poses when
 * cfg->oope == SLAMHIP_OOPE_GMAPPING: poses are scored as ONE call sequence in the given order. */
int slamhip_gm_cache_reset(slamhip_ctx *ctx);
int slamhip_gm_cache_get(slamhip_ctx *ctx, int *cell_xy, double *prob);
int slamhip_gm_cache_set(slamhip_ctx *ctx, const int *cell_xy, double prob);

/* kernel timing of the scoring launches since the last reset (HIP events on the ctx stream) */
int slamhip_profile_enable(slamhip_ctx *ctx, int on);
int slamhip_profile_read(slamhip_ctx *ctx, double *kernel_ms_total, long long *launches,
                         long long *units /* poses x beams launched */, int reset);

/* the same for the map update (K6: count .. apply of one slamhip_map_append_scan or one batched append):
 * recorded HIP events around the pipeline on the context's stream; records = (beam, cell) pairs applied */
int slamhip_profile_read_map_update(slamhip_ctx *ctx, double *ms_total, long long *calls, long long *records,
                                    int reset);

/* ---------------------------------------------------------------- matchers
 * Replace GridScanMatcher::process_scan (src/core/scan_matchers/grid_scan_matcher.h:153-156) of
 *   MonteCarloScanMatcher   (monte_carlo_scan_matcher.h:84-100; enumerator :10-82)
 *   HillClimbingScanMatcher (hill_climbing_scan_matcher.h:128-170; enumerators :10-126)
 *   BruteForceScanMatcher   (brute_force_scan_matcher.h:66-80; enumerator :10-64)
 * through PoseEnumerationScanMatcher::process_scan (pose_enumeration_scan_matcher.h:31-77).
 * The accept/reject chain is evaluated speculatively on the GPU and replayed in the reference's order -- by
 * the next kernel of a device-resident chain (slamhip_matcher_set_device_chain, the default wherever it
 * applies) or, for the configurations the chains do not cover, on the host between batches -- so observers
 * see exactly the reference's event sequence either way. */
typedef struct {
  void *user;
  /* GridScanMatcherObserver (grid_scan_matcher.h:16-32) */
  void (*on_scan_test)(void *user, const double pose[3], double score);
  void (*on_pose_update)(void *user, const double pose[3], double score);
  void (*on_matching_end)(void *user, const double delta[3], double best_score);
} slamhip_observer;

int slamhip_matcher_create_mc(slamhip_ctx *ctx, const slamhip_spe_cfg *cfg, unsigned seed,
                              double translation_dispersion, double rotation_dispersion,
                              unsigned failed_attempts_limit, unsigned attempts_limit,
                              slamhip_matcher **out);
int slamhip_matcher_create_hc(slamhip_ctx *ctx, const slamhip_spe_cfg *cfg,
                              unsigned failed_rounds_limit, double translation_delta,
                              double rotation_delta, slamhip_matcher **out);
int slamhip_matcher_create_bf(slamhip_ctx *ctx, const slamhip_spe_cfg *cfg, const double range9[9],
                              slamhip_matcher **out);
int slamhip_matcher_destroy(slamhip_matcher *m);
/* GridScanMatcher::reset_state (grid_scan_matcher.h:158) */
int slamhip_matcher_reset_state(slamhip_matcher *m);
int slamhip_matcher_set_observer(slamhip_matcher *m, const slamhip_observer *obs);
/* max speculative poses per launch (0 = default) */
int slamhip_matcher_set_batch(slamhip_matcher *m, int max_batch);
/* Hill climbing (1-cell or GMapping OOPE) and Monte Carlo (1-cell OOPE) with device pose trigonometry run their
 * whole accept chain on the device: one process_scan = a chain of kernels with no host in between, each replaying
 * the previous one's speculative candidates in the reference's order (csrc/hc_chain.h, csrc/mc_chain.h;
 * pose_enumeration_scan_matcher.h:31-77, hill_climbing_scan_matcher.h:10-170, monte_carlo_scan_matcher.h:10-100).
 * mode: 0 = the host-driven speculative batches every other configuration uses; 1 = a chain of kernels, one per
 * super-step; 2 (default) = the match as ONE launch whose workgroups stay on the chip and exchange their scores
 * inside it -- hill climbing over the 1-cell and window OOPEs (csrc/hc_resident.hip), Monte Carlo
 * (csrc/mc_resident.hip), and, when asked for explicitly, hill climbing over the GMapping OOPE
 * (csrc/hc_resident_gm.hip; by default a lone GMapping chain stays on mode 1, which is as fast).  Every wait in
 * that launch is bounded, and a match whose workgroups were not all resident (the device was shared) is redone by
 * mode 1, as is everything mode 2 does not cover; threads: workgroup size 256 / 512 / 1024, 0 = default (1024 for
 * hill climbing, 512 for Monte Carlo).  Scores, decisions, observer events and the Monte-Carlo engine's stream are
 * the same bit for bit in every mode. */
int slamhip_matcher_set_device_chain(slamhip_matcher *m, int mode, int threads);
/* mode 2's bookkeeping: matches launched in the co-resident form, and how many of them gave up (bounded wait ran
 * out) and were redone by the chain of kernels */
int slamhip_matcher_resident_stats(slamhip_matcher *m, long long *matches, long long *gave_up);
/* The default mode (SLAMHIP_SUM_TREE256) over the 1-cell AND the window OOPEs (max / mean / overlap: r05) is CHECKED
 * (on = 1, the default) in every form -- co-resident launch, chain of kernels, host-driven batches, brute-force
 * sweep: a `best < candidate` (pose_enumeration_scan_matcher.h:58) between canonical tree sums that lie within 2^-40
 * of each other -- more than the two orders of summation can differ by -- and whose beam terms are not identical
 * (compared through a fingerprint of the term vector) is not decided from the tree sums: the poses in question are
 * summed once more in the reference's beam order (weighted_mean_point_probability_spe.h:108-124) and those sums
 * decide.  The accept chain is then the one SLAMHIP_SUM_SEQUENTIAL gives, at the default mode's speed; reported
 * scores stay the canonical sums.  on = 0: decisions from the tree sums as they are (ties between mathematically
 * equal candidates can then fall the other way: 4 of 200 fuzzed matches over the 1-cell OOPE, 2 of 200 over `max`,
 * tests/test_gpu_hc_chain.py).
 * The GMapping OOPE (r06): a lone matcher's default mode is checked too, by another mechanism -- its per-beam value is
 * exp() of a distance and the fast paths use the device's exp, so there are no beam-order sums of the SAME terms to
 * re-decide from.  A comparison on the walked path whose two scores lie within 2^-40 of each other (two zero scores
 * excepted) is reported before anything has been shown to an observer (device chains: error 7; host-driven batches:
 * MatchJob::gm_unsettled), and the whole match is redone in the EXACT mode: call order, beam-order sums, glibc's exp
 * restated (csrc/libm_exact.h, exact_kernels.hip), with the reference's default RawTrigonometryProvider where the
 * context knows the scan's angles (slamhip_scan_set_angles / slamhip_scan_filter_upload), else the cached provider's
 * angle addition.  Redone matches are counted in slamhip_matcher_chain_stats' steps_rescored.  At the limits GMapping
 * uses (6 failed rounds, init_gmapping.h:58-60) no comparison comes that close and nothing is redone; at limit 27 --
 * steps of 7e-10 m around an optimum -- nearly every match is, and takes the reference's accept path
 * (tests/test_gpu_hc_chain.py).  The filter's per-particle chains (slamhip_gmapping_*) are not checked: they run
 * GMapping's hard-wired limit, and their exact form is SLAMHIP_POSE_TRIG_RAW_EXACT. */
int slamhip_matcher_set_tie_check(slamhip_matcher *m, int on);
/* process_scan on the currently uploaded (filtered) scan; out_delta = best - init */
int slamhip_matcher_process_scan(slamhip_matcher *m, int map_id, const double init_pose[3],
                                 double out_delta[3], double *out_prob);
/* GridScanMatcher::process_scan as the reference declares it (grid_scan_matcher.h:153-156,
 * pose_enumeration_scan_matcher.h:31-77): the RAW scan in -- filter_scan against init_pose (:38), the scan-point weights
 * and the beam trigonometry, the copy to HBM (everything slamhip_scan_filter_upload does, same arguments, gathered in
 * slamhip_raw_scan) -- and the match, in one call.  *kept_n (may be NULL) = points filter_scan kept.  With no point
 * kept the reference's scorer returns 0 / 0 for every candidate and nothing is ever accepted: *out_prob = NaN,
 * out_delta = 0, no launch (observers are not called for such a scan). */
typedef struct slamhip_raw_scan {
  int n;
  const double *range, *angle;
  const int *is_occ;     /* NULL: every point occupied */
  const double *factor;  /* NULL: all 1.0 */
  int trig_mode;         /* SLAMHIP_TRIG_RAW / _CACHED (+ a_min, a_max, a_inc of the cached provider) */
  double a_min, a_max, a_inc;
  unsigned skip_rate;
  double max_range;
  int bounded;           /* see slamhip_scan_filter_upload */
  int weighting;         /* 0 even, 1 viny, 2 ahr */
} slamhip_raw_scan;
int slamhip_matcher_process_raw_scan(slamhip_matcher *m, int map_id, const slamhip_raw_scan *scan,
                                     const double init_pose[3], double out_delta[3], double *out_prob, int *kept_n);
/* K independent matches in shared launches -- PoseEnumerationScanMatcher::process_scan
 * (pose_enumeration_scan_matcher.h:31-77) once per robot / replica (the single-hypothesis matchers do not
 * shard, they replicate): match k = (filtered scan k, initial pose k, map k).  Every match gets the result, the
 * accept trace and the counters of a lone slamhip_matcher_process_scan on the same inputs, bit for bit; what changes
 * is how the GPU is used: a lone hill-climbing match is a chain of ~16 dependent kernels of 253 one-pose workgroups
 * (latency-bound, a quarter of the chip), K of them advance together, one super-step per kernel (grid.y = match,
 * map and scan from a job table in HBM), with trees sized so that all matches' candidates fill the CUs.  Covers
 * what the device chain covers (hill climbing, 1-cell OOPE, default mode, OCC or TBM maps of one cell model);
 * anything else -- and a batch of one -- runs the matches one after another through slamhip_scan_upload +
 * slamhip_matcher_process_scan (the context's uploaded scan is unspecified after the call).  The matcher's
 * observer, if set, sees the matches' event sequences one after another, job 0 first, each closed by
 * on_matching_end.  out_deltas: 3 doubles per job; out_probs: one. */
typedef struct {
  int map_id;
  int scan_slot;                            /* >= 0: the scan stored in that slot (slamhip_scan_store), resident in
                                             * HBM -- n and the arrays below are ignored; -1: the arrays below */
  int n;                                    /* beams of the filtered scan, arrays as for slamhip_scan_upload */
  const double *range, *cos_a, *sin_a, *weight;
  const double *factor;                     /* may be NULL = all 1.0 */
  double init_pose[3];
} slamhip_match_job;
int slamhip_matcher_process_scan_batch(slamhip_matcher *m, int n_jobs, const slamhip_match_job *jobs,
                                       double *out_deltas, double *out_probs);
/* counters of job `job` of the last batch (slamhip_matcher_stats gives the sums; its `launches` the longest
 * chain's super-steps); on_device_chain: 1 when the match ran in the shared launches */
int slamhip_matcher_batch_stats(slamhip_matcher *m, int job, long long *scorer_calls, long long *poses_evaluated,
                                long long *super_steps, int *on_device_chain);
/* counters of the last process_scan: scorer calls the reference would have made
 * (= on_scan_test events), poses actually evaluated on the GPU, launches */
int slamhip_matcher_stats(slamhip_matcher *m, long long *scorer_calls, long long *poses_evaluated,
                          long long *launches);

/* device chain only (zeros otherwise), last process_scan: kernels launched (super-steps + run-ahead launches that
 * found the chain finished) and super-steps the checked default mode scored a second time in beam order because
 * a comparison on the path was too close for the canonical tree sum to settle */
int slamhip_matcher_chain_stats(slamhip_matcher *m, long long *kernels_launched, long long *steps_rescored);
/* last process_scan / batch: scorer calls (of slamhip_matcher_stats' count) that were reported in closed form instead
 * of being scored -- the tail of a hill-climbing match whose candidates have all become the best pose itself
 * (SLAMHIP_OPT_INERT_TAIL; hill_climbing_scan_matcher.h:83-101).  0 on every other path. */
int slamhip_matcher_tail_stats(slamhip_matcher *m, long long *calls_closed_form);

/* host-side time split of the last process_scan in microseconds: building the speculation DAG,
 * staging poses, launch + wait for scores, replay of the accept chain */
int slamhip_matcher_timing(slamhip_matcher *m, double *build_us, double *stage_us, double *score_us,
                           double *replay_us);

/* ---------------------------------------------------------------- particle filter (K4/K5)
 * ParticleFilter::normalize_weights / UniformResamling (src/core/particle_filter.h:34-66,108-121)
 * in the reference's summation order (host; N <= a few hundred).  Sharded runs all-gather the raw
 * weights first (RCCL) and call these on every rank so indices are bit-identical. */
int slamhip_pf_normalize(int n, double *weights);
int slamhip_pf_resampling_is_required(int n, const double *weights, int *required);
int slamhip_pf_resample(int n, const double *weights, uint32_t seed, unsigned *out_idx);
int slamhip_pf_heaviest(int n, const double *weights, int *index);

/* ---------------------------------------------------------------- GMapping particle filter
 * Replaces LaserScanGridWorld::handle_sensor_data of GmappingParticleFilter
 * (src/slams/gmapping/gmapping_particle_filter.h:45-50; per particle GmappingWorld::update_robot_pose
 * / handle_observation, src/slams/gmapping/gmapping_world.h:57-101) for the likelihood part of
 * the step: odometry, matching gate, pose noise, HC(6, 0.1, 0.1) scan matching of all particles in
 * lock-step on the GPU, weight update, normalisation, N_eff test, multinomial resampling with
 * duplicated particles and master hand-over.  The map update inside the step (gmapping_world.h:93-97) comes in
 * two forms: slamhip_gmapping_set_map_update (the reference's ONE shared map, updated particle after particle)
 * and slamhip_gmapping_enable_particle_maps (a copy-on-write map per particle, one batched update per step).
 *
 * A filter object holds the particles [first, first + count) of n_total (one object per GPU when
 * particles are sharded).  Sharded step:
 *   1. every rank:  slamhip_gmapping_predict_match(...)            -> raw weights of its shard
 *   2. caller:      all-gather the raw weights (n_total doubles)      [the one collective, RCCL]
 *   3. every rank:  slamhip_gmapping_plan_resample(all weights)    -> identical decision + indices
 *   4. if required: slamhip_gmapping_export -> all-gather n_total records -> slamhip_gmapping_import
 * slamhip_gmapping_step does 1-4 for an unsharded filter. */
typedef struct slamhip_gmapping slamhip_gmapping;
typedef struct {
  /* GMappingParams (src/slams/gmapping/gmapping_world.h:16-34, defaults init_gmapping.h:15-34) */
  double mean_sample_xy, sigma_sample_xy, mean_sample_th, sigma_sample_th;
  double min_sm_lim_xy, max_sm_lim_xy, min_sm_lim_th, max_sm_lim_th;
  /* HillClimbingScanMatcher(6, 0.1, 0.1) hard-wired in init_gmapping.h:58-60 */
  unsigned hc_failed_rounds_limit;
  double hc_translation, hc_rotation;
  /* WeightedMeanPointProbabilitySPE(oope, EvenSPW, skip_rate, max_range), init_scan_matching.h:94-109 */
  unsigned sp_skip_rate;
  double sp_max_usable_range;
  /* GmappingOccupancyObservationPE(fullness_th, window), init_gmapping.h:36-45 */
  double oope_fullness_th;
  int oope_window;
  int pose_trig; /* SLAMHIP_POSE_TRIG_* */
} slamhip_gmapping_params;

/* seeds: one per LOCAL particle = what std::random_device hands the GmappingWorld ctor
 * (gmapping_world.h:51).  ctx may be NULL for host-only bookkeeping (steps 3-4). */
int slamhip_gmapping_create(slamhip_ctx *ctx, const slamhip_gmapping_params *prm, int n_total,
                            int first, int count, const uint32_t *seeds, slamhip_gmapping **out);
int slamhip_gmapping_destroy(slamhip_gmapping *g);
/* raw scan (unfiltered); raw_weights_out: count doubles (weight * scan probability, unnormalised) */
int slamhip_gmapping_predict_match(slamhip_gmapping *g, int map_id, int n_raw, const double *range,
                                   const double *angle, const int *is_occ, const double odom_delta[3],
                                   double *raw_weights_out);
/* ParticleFilter::normalize_weights + the try_resample gates (gmapping_particle_filter.h:88-99,
 * particle_filter.h:34-43) + UniformResamling::resample (:45-66) over ALL n_total weights;
 * idx_out (n_total) is written when *required becomes 1 */
int slamhip_gmapping_plan_resample(slamhip_gmapping *g, const double *all_raw_weights,
                                   uint32_t resample_seed, int *required, unsigned *idx_out);
size_t slamhip_gmapping_blob_size(void);
int slamhip_gmapping_export(slamhip_gmapping *g, void *blobs_out /* count records */);
/* ParticleFilter::try_resample body (particle_filter.h:88-105) + ensure_master_exists */
int slamhip_gmapping_import(slamhip_gmapping *g, const void *all_blobs /* n_total records */,
                            const unsigned *idx /* n_total */);
int slamhip_gmapping_step(slamhip_gmapping *g, int map_id, int n_raw, const double *range,
                          const double *angle, const int *is_occ, const double odom_delta[3],
                          uint32_t resample_seed, int *resampled, unsigned *idx_out);
int slamhip_gmapping_set(slamhip_gmapping *g, const double *poses, const double *weights);
/* cfg != NULL: every matching particle appends its scan to the map (scan adder of init_gmapping,
 * init_occupancy_mapping.h:82-92; rule and scan_quality are overridden: GmappingBaseCell, 1.0) right
 * after its match, BEFORE the next particle matches -- the reference's particles share one map
 * object (Q20), which makes this step sequential and unshardable.  NULL switches it off. */
int slamhip_gmapping_set_map_update(slamhip_gmapping *g, const slamhip_scan_adder_cfg *cfg);
/* Per-particle maps: every particle gets its OWN map, a copy-on-write copy of the bound
 * GMAPPING window `map_id`, held as 128x128-cell tiles in a device pool with the sharing semantics of
 * UnboundedLazyTiledGridMap (src/core/maps/lazy_tiled_grid_map.h:18-187: a map copy copies tile
 * references :40-45, untouched area is one shared unknown tile :28-34, a write clones a shared tile
 * first :57-71).  This is what GmappingWorld's per-particle map member is for
 * (gmapping_world.h:36-55); the reference revision never separates them (Q20), so the mode has no
 * reference run -- parity is against the oracle with per-particle maps.  The step then keeps the
 * lock-step matching and appends the scans of all matched particles in ONE batched K6; a resampling
 * copies tile tables (particle_filter.h:92-96).  extent_tiles: side of the initial virtual extent in
 * tiles; it grows by whole tiles when a scan reaches beyond it, like the reference's unbounded maps
 * (lazy_tiled_grid_map.h:128-187; cells outside read as unknown); pool_tiles: capacity (768 KiB each).
 * A shard [first, first+count) of the filter holds the maps of its own particles; when a resampling
 * draws a particle that lives on another rank its map travels as one exported buffer (below). */
int slamhip_gmapping_enable_particle_maps(slamhip_gmapping *g, int map_id, const slamhip_scan_adder_cfg *cfg,
                                          int extent_tiles, int pool_tiles);
/* external window of one particle's map: payload3 = (prob_occ, obstacle x, obstacle y) per cell,
 * aux2 = (hits, tries) per cell; either may be NULL */
int slamhip_gmapping_particle_map_download(slamhip_gmapping *g, int particle, int x0, int y0, int w, int h,
                                           double *payload3, double *aux2);
/* The append half of GmappingWorld::handle_observation (gmapping_world.h:93-97) for a set of local
 * particles at once: the raw scan is appended to the map of particle particles[k] from poses3[3k..3k+2] --
 * ONE batched K6 over all (particle, beam) pairs, copy-on-write first.  The filter step does this itself for
 * the particles that matched; this entry point is for callers that place scans from their own poses. */
int slamhip_gmapping_particle_maps_append(slamhip_gmapping *g, int n_jobs, const int *particles,
                                          const double *poses3, int n_raw, const double *range,
                                          const double *angle, const int *is_occ, long long *n_updates);
int slamhip_gmapping_particle_map_stats(slamhip_gmapping *g, long long *tiles_in_use, long long *tiles_shared,
                                        long long *bytes, long long *cow_copies, long long *cell_updates);
/* Migration of a particle's map to another rank (the "moving a duplicated particle to another GPU"
 * step of SURVEY 8e).  Export: every tile the LOCAL particle references (except the unknown tile) into a
 * host buffer: int64 n_tiles, per tile four int32 (x, y of its first cell in external coordinates -- so
 * that pools whose extents grew differently still agree --, ordinal of the common ancestor tile it
 * still is or -1, 0), then per non-ancestor tile 16384 x 4 payload and 16384 x 2 counter doubles.
 * The receiving pool grows first if the map reaches beyond it.  Exports must be taken before any rank
 * imports. */
int slamhip_gmapping_particle_map_export_size(slamhip_gmapping *g, int particle, size_t *bytes);
int slamhip_gmapping_particle_map_export(slamhip_gmapping *g, int particle, void *host_buf, size_t cap);
/* slamhip_gmapping_import for a filter with per-particle maps: new local particle l takes the map of
 * old particle idx[first + l] -- a table copy when that one is local, otherwise the exported buffer
 * remote_bufs[k] with remote_src[k] == idx[first + l] (GLOBAL particle index; a source used by several
 * new particles is listed and imported once, its tiles are then shared copy-on-write). */
int slamhip_gmapping_import_maps(slamhip_gmapping *g, const void *all_blobs, const unsigned *idx, int n_remote,
                                 const int *remote_src, const void *const *remote_bufs);
int slamhip_gmapping_get(slamhip_gmapping *g, double *poses, double *weights, int *is_master);

/* ---------------------------------------------------------------- sharding over GPUs (RCCL over xGMI)
 * What gets distributed is ParticleFilter's particle loop (src/core/particle_filter.h:108-112 over
 * GmappingWorld::handle_sensor_data) -- one shard [first, first + count) of the particles per GPU, one
 * process (or thread) per GPU, one context each -- and what has to be collected is what
 * normalize_weights / UniformResamling read: ALL raw weights, in particle order (:34-66).  The one
 * collective of a step is therefore an all-gather of n_total doubles; an all-reduce of (sum w, sum w^2)
 * would be smaller but adds in another order than the reference does, and resampling indices must stay
 * bit-exact.  RCCL is loaded when the first of these functions is called (librccl.so is not a link
 * dependency of libslamhip.so).
 *   rank 0:      slamhip_shard_unique_id(id)          -> hand the 128 bytes to the other ranks (MPI, a
 *                                                        file, a socket: whatever started the processes)
 *   every rank:  slamhip_shard_init(ctx, rank, world, id)
 *   per scan:    slamhip_gmapping_step_sharded(...)   = match_begin, the carry exchange, match_finish,
 *                                                        all-gather of the raw weights, plan_resample and,
 *                                                        when a resampling happens, all-gather of the
 *                                                        particle records + import
 *
 * slamhip_gmapping_step_sharded needs ONE collective per step in the common case: the carry records travel together
 * with the raw weights the particles will have if no cache hand-over needs repair, and whether one does is read off
 * the records by every rank alike (no flags exchanged).
 *
 * A caller with its own means of moving bytes between the ranks (MPI, shared memory, a test harness that runs the
 * ranks as threads of one process) joins a group with slamhip_shard_attach instead of slamhip_shard_init and hands in
 * a slamhip_shard_transport; everything above the transport -- slamhip_shard_allgather's padding,
 * slamhip_gmapping_step_sharded -- is the same code either way (tests/native/loopback_transport.cpp is such a
 * transport: it is how the sharded step runs with world > 1 on the one GPU of a test box, RCCL admitting one rank
 * per device). */
#define SLAMHIP_SHARD_ID_BYTES 128
int slamhip_shard_unique_id(void *id_out);
int slamhip_shard_init(slamhip_ctx *ctx, int rank, int world, const void *id);
int slamhip_shard_destroy(slamhip_ctx *ctx);
int slamhip_shard_info(slamhip_ctx *ctx, int *rank, int *world);
/* all-gather of per-rank blocks: rank r contributes counts[r] elements of elem_bytes bytes (host memory,
 * `local`); `all_out` (host) receives the blocks in rank order.  Blocks are padded to the largest one on
 * the wire (100 particles on 8 GPUs are 13 x 4 + 12 x 4). */
int slamhip_shard_allgather(slamhip_ctx *ctx, const void *local, const int *counts, int elem_bytes,
                            void *all_out);
/* bytes moved through RCCL and collectives issued since slamhip_shard_init */
int slamhip_shard_stats(slamhip_ctx *ctx, long long *collectives, long long *bytes);
/* Point to point, all ranks together: this rank sends n_send blocks and receives n_recv.  Buffers are DEVICE memory of
 * the context's GPU; messages between one pair of ranks are matched in the order they are listed on both sides.  Over
 * RCCL this is one ncclGroupStart .. ncclSend / ncclRecv .. ncclGroupEnd on the context's stream (xGMI is point to
 * point: a pair's bytes travel on the link between the two GPUs), awaited before the call returns.  It is what carries a
 * resampled particle's map to the rank that drew it (particle_filter.h:88-103: `*new_particle = *sampled` copies
 * the whole world; lazy_tiled_grid_map.h:40-71: tile by tile). */
typedef struct {
  int peer;
  void *buf;
  size_t bytes;
} slamhip_shard_msg;
int slamhip_shard_exchange(slamhip_ctx *ctx, int n_send, const slamhip_shard_msg *send, int n_recv,
                           const slamhip_shard_msg *recv);
int slamhip_shard_p2p_stats(slamhip_ctx *ctx, long long *exchanges, long long *bytes_sent);
/* Every wait on a collective of the built-in (RCCL) transport is bounded: slamhip_shard_allgather / _exchange poll the
 * stream until the deadline (default 30 s; ms <= 0 restores it) and then return SLAMHIP_ERR_TIMEOUT -- a peer that
 * died INSIDE a collective, which no status word can report.  The communicator is aborted (ncclCommAbort) and the
 * group is broken from then on: every later collective of this context fails at once with SLAMHIP_ERR_STATE, so a
 * sharded step in flight is abandoned on every survivor within the deadline; slamhip_shard_destroy + a fresh
 * slamhip_shard_init among the survivors start over.  (An attached transport bounds its own waits and reports
 * through its return code.) */
int slamhip_shard_set_timeout(slamhip_ctx *ctx, int ms);
/* A transport of the caller's: allgather moves equal blocks of host memory (every rank's block_bytes from send_host
 * into recv_host, world x block_bytes in rank order), exchange has the contract of slamhip_shard_exchange (device
 * buffers; the context's stream is idle when it is called), destroy (may be NULL) runs when the context leaves the
 * group.  Each returns 0 on success.  The table is copied. */
typedef struct {
  void *user;
  int (*allgather)(void *user, const void *send_host, size_t block_bytes, void *recv_host);
  int (*exchange)(void *user, int n_send, const slamhip_shard_msg *send, int n_recv, const slamhip_shard_msg *recv);
  void (*destroy)(void *user);
} slamhip_shard_transport;
int slamhip_shard_attach(slamhip_ctx *ctx, int rank, int world, const slamhip_shard_transport *t);

/* A step in phases, for callers that bring their own collective: match_begin = odometry, gate, pose noise
 * and the lock-step matching of the local shard; match_finish = poses, weights (and the batched map
 * update of per-particle maps); slamhip_gmapping_predict_match is begin + finish.
 * Between them, sharded steps repair the ONE OOPE cache the reference's particles hand from one to the
 * next (gmapping_occupancy_observation_pe.h:21-24,36-37,43-44): every shard starts its
 * first job without a carry (slamhip_gmapping_set_shard_chain), publishes a slamhip_carry_record, and
 * checks its first job against the final cache entry of the shard before it -- re-matching on a hit --
 * until no record changes any more (at most `world` rounds; none in practice). */
typedef struct {
  int has_active;                 /* some particle of the shard matched in this step */
  int first_cx, first_cy;         /* end-point cell of the first beam of its first scored pose ... */
  double first_v0;                /* ... and the fresh value of that beam */
  int carry_cx, carry_cy;         /* the cache entry the shard's last job ended with */
  double carry_prob;
} slamhip_carry_record;
int slamhip_gmapping_set_shard_chain(slamhip_gmapping *g, int on);
int slamhip_gmapping_match_begin(slamhip_gmapping *g, int map_id, int n_raw, const double *range,
                                 const double *angle, const int *is_occ, const double odom_delta[3]);
int slamhip_gmapping_carry_record(slamhip_gmapping *g, slamhip_carry_record *rec);
int slamhip_gmapping_carry_fix(slamhip_gmapping *g, const slamhip_carry_record *all, int world, int rank,
                               int *changed);
int slamhip_gmapping_carry_commit(slamhip_gmapping *g, const slamhip_carry_record *all, int world);
int slamhip_gmapping_match_finish(slamhip_gmapping *g, double *raw_weights_out);
/* One scan on a shard, collectives included (the filter's context has joined a group: slamhip_shard_init or
 * slamhip_shard_attach).  resampled / idx_out as in slamhip_gmapping_step; idx_out holds n_total indices.
 * Filters with per-particle maps (slamhip_gmapping_enable_particle_maps) are covered: when a resampling draws a
 * particle that lives on another rank, its map travels inside this call -- every rank reads the same migration plan
 * off the resampling indices; two small all-gathers carry the sizes and the maps' headers (tile positions, ancestor
 * ordinals), ONE slamhip_shard_exchange carries the tile contents device to device (RCCL send / recv), then
 * slamhip_gmapping_import_maps' work is done from the received buffers (particle_filter.h:88-103: the reference's
 * resampling copies whole worlds; lazy_tiled_grid_map.h:40-71: maps copy tile by tile).
 * Errors: a rank that fails inside a step says so in the status word of the step's next collective, and EVERY rank
 * returns from that step with an error (the failing one with its own code, the others with SLAMHIP_ERR_STATE)
 * instead of waiting for it; a failure behind a step's last collective is announced at the start of the next
 * step.  A filter whose step was abandoned is usable again (slamhip_gmapping_match_abort does the same for callers
 * that drive the phases themselves): the particles keep the odometry and pose noise the step had applied. */
int slamhip_gmapping_step_sharded(slamhip_gmapping *g, int map_id, int n_raw, const double *range,
                                  const double *angle, const int *is_occ, const double odom_delta[3],
                                  uint32_t resample_seed, int *resampled, unsigned *idx_out);
int slamhip_gmapping_match_abort(slamhip_gmapping *g);
/* maps this shard received from other ranks and tile bytes it sent, since the filter was created */
int slamhip_gmapping_migration_stats(slamhip_gmapping *g, long long *maps_received, long long *tile_bytes_sent);
int slamhip_gmapping_stats(slamhip_gmapping *g, long long *scorer_calls, long long *poses_evaluated,
                           long long *launches, long long *carry_reruns);

#ifdef __cplusplus
}
#endif
#endif /* SLAMHIP_H */
