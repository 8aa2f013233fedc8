/* oracle/slam_oracle.h -- TEST INFRASTRUCTURE ONLY (see slam_oracle.c header).
 *
 * Plain-C restatement of the slam-constructor scan-matching / particle-filter hot path.
 * Every function cites the reference file:line it restates (paths relative to the
 * reference root, slam_constructor 0.9.1).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library.
 */
#ifndef SLAM_ORACLE_H
#define SLAM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* cell payload models (same values as include/slamhip.h) */
enum { ORC_CELL_OCC = 0, ORC_CELL_TBM = 1, ORC_CELL_GMAPPING = 2 };
enum { ORC_OOPE_OBSTACLE = 0, ORC_OOPE_MAX = 1, ORC_OOPE_MEAN = 2, ORC_OOPE_OVERLAP = 3,
       ORC_OOPE_GMAPPING = 4 };
enum { ORC_OIE_DISCREPANCY = 0, ORC_OIE_OCCUPANCY = 1 };
enum { ORC_TRIG_RAW = 0, ORC_TRIG_CACHED = 1 };
enum { ORC_SUM_SEQUENTIAL = 0, ORC_SUM_TREE256 = 1 };
enum { ORC_SM_MC = 0, ORC_SM_HC = 1, ORC_SM_BF = 2 };

/* Dense window of a GridMap: internal cell (ix,iy) = external + origin; payload is
 * row-major [height][width][stride], stride = 1 (OCC: prob_occ), 4 (TBM: u,e,o,c),
 * 3 (GMAPPING: prob_occ, obst.x, obst.y). */
typedef struct {
  int cell_model;
  int width, height;
  int origin_x, origin_y;
  double scale;
  const double *payload;
  double unknown[4]; /* payload of the prototype / out-of-window cell */
  int bounded;       /* 1: has_cell() tests the window (PlainGridMap); 0: Unbounded* */
} orc_map;

typedef struct {
  int n;
  const double *range, *angle;   /* polar scan points */
  const double *weight, *factor; /* per-point weight (ScanPointWeighting) and factor */
  int trig_mode;
  double a_min, a_delta;         /* cached provider: index = round((a - a_min)/a_delta) */
  int table_n;
  const double *tab_sin, *tab_cos;
} orc_scan;

typedef struct {
  int oope, oie;
  double area[4];        /* SPEParams::sp_analysis_area (bot, top, left, right) */
  double gm_fullness_th; /* GmappingOccupancyObservationPE */
  int gm_window;
  int sum_order;
} orc_spe_cfg;

typedef struct { int cx, cy; double prob; } orc_gm_cache; /* init: {0,0,-1} */

/* ---- RNG (libstdc++ <random> restated) ---- */
typedef struct { uint32_t mt[624]; int idx; } orc_mt19937;
void orc_mt_seed(orc_mt19937 *g, uint32_t seed);
uint32_t orc_mt_next(orc_mt19937 *g);
double orc_canonical(orc_mt19937 *g);
typedef struct { double mean, stddev, saved; int has_saved; } orc_normal;
void orc_normal_init(orc_normal *d, double mean, double stddev);
double orc_normal_sample(orc_normal *d, orc_mt19937 *g);
double orc_uniform_real(orc_mt19937 *g, double a, double b);

/* ---- trig table, filter, weights ---- */
int orc_build_trig_table(double a_min, double a_max, double a_inc, double *sin_out,
                         double *cos_out, int cap);
int orc_filter_scan(const orc_map *map, int n, const double *range, const double *angle,
                    const int *is_occ, const double *pose, unsigned skip_rate, double max_range,
                    const orc_scan *trig, int *kept_idx);
void orc_weights_even(int n, double *out);
void orc_weights_viny(int n, const double *range, const double *angle, double *out);
void orc_weights_ahr(int n, const double *range, const double *angle, double *out);

/* ---- scoring ---- */
void orc_endpoint(const orc_scan *scan, int i, const double *pose, double sin_b, double cos_b,
                  double *wx, double *wy);
double orc_oope_probability(const orc_map *map, const orc_spe_cfg *cfg, double ox, double oy,
                            const double *area4, orc_gm_cache *cache);
void orc_score_poses(const orc_map *map, const orc_scan *scan, const orc_spe_cfg *cfg, int n_poses,
                     const double *poses, double *scores, orc_gm_cache *cache);

/* ---- pose enumerators + accept loop ---- */
typedef struct {
  int kind;
  /* MC (monte_carlo_scan_matcher.h:10-82) */
  orc_mt19937 eng;
  orc_normal rv[3];
  unsigned max_failed, max_poses, failed, poses_nm;
  double base_td, base_rd, td, rd;
  /* HC (hill_climbing_scan_matcher.h:10-126) */
  unsigned max_failed_rounds, failed_rounds;
  double base_dt, base_dr, dt, dr;
  unsigned action_id;
  int base_set, round_failed;
  double base_pose[3];
  /* BF (brute_force_scan_matcher.h:10-64) */
  double bf[9], bx, by, bt;
  int bf_base_set;
} orc_enumerator;

void orc_enum_init_mc(orc_enumerator *e, unsigned seed, double sigma_t, double sigma_r,
                      unsigned max_failed, unsigned max_poses);
void orc_enum_init_hc(orc_enumerator *e, unsigned max_failed_rounds, double dt, double dr);
void orc_enum_init_bf(orc_enumerator *e, const double *p9);
void orc_enum_reset(orc_enumerator *e);
int orc_enum_has_next(const orc_enumerator *e);
void orc_enum_next(orc_enumerator *e, const double *best_pose, double *out_pose);
void orc_enum_feedback(orc_enumerator *e, int ok);

/* PoseEnumerationScanMatcher::process_scan on an already *filtered* scan.
 * Returns number of scorer calls; trace arrays may be NULL; res = {prob, dx, dy, dth}. */
int orc_process_scan(orc_enumerator *e, const orc_map *map, const orc_scan *scan,
                     const orc_spe_cfg *cfg, const double *init_pose, double *res, int cap,
                     double *tr_poses, double *tr_scores, int *tr_accepted, orc_gm_cache *cache);

/* ---- particle filter ---- */
void orc_normalize_weights(int n, double *w);
int orc_resampling_is_required(int n, const double *w);
void orc_resample(int n, const double *w, uint32_t seed, unsigned *out_idx);
int orc_heaviest(int n, const double *w);

/* ---- GMapping particle filter (src/slams/gmapping/gmapping_world.h, gmapping_particle_filter.h,
 *      src/core/particle_filter.h) WITHOUT the map update (the reference run it is pinned against
 *      uses slam/mapping/max_range = 0, which turns append_scan into a no-op) ---- */
typedef struct orc_adder_s {
  double base4[4]; /* occupied prob, qual, empty prob, qual */
  double blur, max_range;
  int est_kind;    /* 0 const, 1 area */
  double shift_amount;
} orc_adder;

typedef struct {
  double pose[3], raw_odom[3], weight;
  int is_master, scan_is_first;
  orc_mt19937 eng;
  orc_normal guess[3];          /* _pose_guess_rv */
  int nsd_is_normal;            /* _next_sm_delta_rv: 0 = UniformRV1D(a,b), 1 = GaussianRV1D(0,0) */
  double nsd_a[3], nsd_b[3];
  orc_normal nsd_norm[3];
  double dsl[3], nsd[3];        /* _delta_since_last_sm, _next_sm_delta */
} orc_particle;

typedef struct {
  int n;
  orc_particle *p;
  double traversed[3];          /* _traversed_since_last_resample */
  double gp[8];                 /* GMappingParams ctor arguments */
  unsigned hc_limit; double hc_dt, hc_dr;
  unsigned skip_rate; double max_range;
  orc_spe_cfg cfg;
  orc_gm_cache cache;           /* ONE OOPE shared by all particles (Q20) */
  /* optional map update inside the step (gmapping_world.h:93-97): scan adder parameters and the
   * MUTABLE payload / (hits, tries) arrays of the window the `map` argument of the step views */
  const struct orc_adder_s *upd;
  double *upd_payload, *upd_aux;
  /* per-particle maps (what GmappingWorld's own map member is for; the reference revision shares
   * one object, Q20): particle i reads and updates pm_payload[i] / pm_aux[i], windows of the geometry
   * the step's `map` argument describes; a resampling copies them (particle_filter.h:92-96) */
  double **pm_payload, **pm_aux;
  size_t pm_payload_doubles, pm_aux_doubles;
  long long scorer_calls;       /* of the last step */
} orc_gmapping;

orc_gmapping *orc_gmapping_create(int n, const double *gp8, const uint32_t *seeds, unsigned hc_limit,
                                  double hc_dt, double hc_dr, unsigned skip_rate, double max_range);
void orc_gmapping_destroy(orc_gmapping *g);
/* one handle_sensor_data; returns 1 if resampling happened; idx_out (n) receives the resampling
 * indices when it did.  extra_seeds feed the GmappingWorld ctor of every duplicated particle. */
int orc_gmapping_step(orc_gmapping *g, const orc_map *map, int n_raw, const double *range,
                      const double *angle, const int *is_occ, const double *odom_delta,
                      uint32_t resample_seed, int n_extra, const uint32_t *extra_seeds,
                      unsigned *idx_out);
void orc_gmapping_get(const orc_gmapping *g, double *poses, double *weights, int *is_master);
long long orc_gmapping_scorer_calls(const orc_gmapping *g);
void orc_gmapping_set_update(orc_gmapping *g, const orc_adder *upd, double *payload, double *aux);
/* every particle gets its own copy of (payload, aux); updates inside the step go to the copies */
void orc_gmapping_set_particle_maps(orc_gmapping *g, const orc_adder *upd, const double *payload,
                                    size_t payload_doubles, const double *aux, size_t aux_doubles);
/* GridMapScanAdder::append_scan on particle `particle`'s own map from `pose` (trig NULL = raw provider) */
long long orc_gmapping_particle_map_append(orc_gmapping *g, const orc_map *map, int particle, const double *pose,
                                           int n_raw, const double *range, const double *angle, const int *is_occ,
                                           const orc_scan *trig);
void orc_gmapping_copy_particle_map(const orc_gmapping *g, int particle, double *payload_out, double *aux_out);

/* ---- map update (map_update_oracle.c) ---- */
/* cell update rules = the reference's GridCell subclasses' operator+= */
enum { ORC_RULE_LAST = 0, ORC_RULE_AFFINE = 1, ORC_RULE_MEAN = 2, ORC_RULE_TBM = 3, ORC_RULE_GMAPPING = 4 };
/* GridMapScanAdder::append_scan with the const occupancy estimator on a mutable window.
 * payload: [h][w][stride] of map->cell_model; aux: MEAN -> n [h][w], GMAPPING -> (hits, tries)
 * [h][w][2], else NULL; base4 = {occ prob, occ qual, empty prob, empty qual}.  Returns the number
 * of cell updates, -1 if a touched cell lies outside the window. */
long long orc_append_scan(const orc_map *map, double *payload, double *aux, int rule, const double *pose,
                          int n, const double *range, const double *angle, const int *is_occ,
                          const orc_scan *trig, double scan_quality, const double *base4, double blur,
                          double max_range);
/* est_kind 0 = ConstOccupancyEstimator, 1 = AreaOccupancyEstimator (oracle/area_estimator.h);
 * shift_amount = the function-local static of ensure_segment_not_on_edge (Q27) */
long long orc_append_scan_ex(const orc_map *map, double *payload, double *aux, int rule, const double *pose,
                             int n, const double *range, const double *angle, const int *is_occ,
                             const orc_scan *trig, double scan_quality, const double *base4, double blur,
                             double max_range, int est_kind, double shift_amount);
/* per-point observation quality (grid_map_scan_adders.h:24-43): kind 0 idle, 1 angle-histogram reciprocal */
void orc_omqe_quality(int kind, int n, const double *range, const double *angle, double *out);
/* append_scan with a per-point quality factor (NULL = IdleOMQE) */
long long orc_append_scan_q(const orc_map *map, double *payload, double *aux, int rule, const double *pose,
                            int n, const double *range, const double *angle, const int *is_occ,
                            const orc_scan *trig, double scan_quality, const double *base4, double blur,
                            double max_range, int est_kind, double shift_amount, const double *beam_quality);
/* pieces exposed for the known answers of the reference's own unit tests
 * (tests/golden/make_golden_ref_tests.py -> reference_test_vectors.json) */
int orc_discrete_segment(int bx, int by, int ex, int ey, int cap, int *out_xy);
double orc_ah_angle(double base_x, double base_y, double sp_x, double sp_y);
void orc_area_estimate(const double *beam4, const double *cell4, int is_occ, const double *base4,
                       double low_qual, double unknown_qual, double *out2);
int orc_world_to_cells(double scale, double x0, double y0, double x1, double y1, int cap,
                       int *out_xy);

#ifdef __cplusplus
}
#endif
#endif
