// map_update.hip -- K6: the map update of one scan on the GPU, in the reference's update order.
//
// Restates (paths relative to the reference root):
//   GridMapScanAdder::append_scan                     src/core/maps/grid_map_scan_adders.h:54-75
//   WallDistanceBlurringScanAdder::handle_scan_point  :138-172, blur_cell_dist :176-189
//   RegularSquaresGrid::world_to_cells                src/core/maps/regular_squares_grid.h:56-101
//   DiscreteSegment2D (Bresenham fail-over)           src/core/geometry_discrete_primitives.h:55-104
//   ConstOccupancyEstimator                           src/core/maps/const_occupancy_estimator.h:6-17
//   cell updates: GridCell (grid_cell.h:27-30), AffineQualityMergeCell / MeanProbabilityCell
//     (naive_grid_cells.h:14-20,33-40), TbmBaseCell (tbm_grid_cells.h:12-19,57-66;
//     transferable_belief_model.h:102-143), GmappingBaseCell (src/slams/gmapping/gmapping_grid_cell.h:20-33)
//
// The cell update is order dependent (running means, TBM conjunction + normalisation) and the beams
// of one scan overlap near the robot, so atomics cannot reproduce the sequential result.
// Exact scheme (has the long form and the measurements):
//   1. per beam: endpoint, range gate, number of cells |dx|+|dy|+1, the per-beam constants of the observation
//      (MuBeam) -- k_mu_count in a batch; in a plain call k_mu_emit does it itself and the host, which walks the
//      beams anyway to size the buffers, leaves every beam's first record slot in pinned memory
//   2. k_mu_emit    ONE WAVE per beam, one lane per step: the cells of the 4-connected walk in closed form, every
//                   step verified against the recurrence's own decision; ties, axis-parallel beams and walks
//                   that rounding sends astray fall back to the sequential walk (fuzzy tie rule, Bresenham
//                   fail-over).  It leaves the sort key and the beam of every visited cell, beam-major
//   3. grouping by cell, stable in beam order -- within a beam a cell is visited once, so beam order IS the
//      reference's update order for that cell.  Plain call: a counting sort over the window around the scan
//      (per-cell bins filled by k_mu_emit, exclusive scan, k_mu_scatter, k_mu_rank; the cells next to the robot
//      keep a bitmap of their beams instead of a bin, k_mu_near_bits); batch, or a window too large for bins:
//      rocprim::radix_sort_pairs (stable) of (cell key, beam)
//   4. the observation (occupancy estimate, blur) of every (beam, cell) pair, 8 bytes per record, TBM cells 16:
//      k_mu_rank in a plain call, k_mu_gather behind the radix sort
//   5. k_mu_apply   one thread per distinct cell applies its records sequentially; chains of >= 64
//                   records are then streamed through the whole wave that holds their head
//   batch of GMapping-cell maps: 3-5 run only over the records that NEED an order.  A valid free observation of a
//   cell whose mean is 0 (or that was never observed) only counts a try -- such updates commute -- so k_mu_classify
//   settles them with one f64 atomic each and hands the rest (cells a beam may hit or blur, cells hit before:
//   a few per cent) to the sort (mu_batch_fast_tail)
// HBM traffic: per (beam, cell) an 8-byte (key, beam) pair written, sorted and read, one 8-byte
// observation written and read, plus one read-modify-write of the cell (8-48 bytes) per distinct cell; on the
// batch's fast path a 4-byte key written and read, the cell's mean read and its try counter updated in place.
// The kernels live in map_update_kernels.h; this file holds the two host drivers (one scan into a bound
// dense map; one scan from many poses into the copy-on-write maps of a particle filter).

#include <string.h>  // rocprim's texture iterator calls ::memset without including it

#include <climits>
#include <cmath>
#include <cstring>
#include <limits>
#include <type_traits>

#include <rocprim/rocprim.hpp>

#include "slamhip_internal.h"
#include "area_estimator_device.h"

namespace slamhip {

// one scan appended from one pose into one map slot; a batch appends the SAME scan from many poses
// (the particles of the filter), each into its own copy-on-write map (tile_pool.h)
struct MuJob {
  double px, py, sn, cs;
  int slot, pad;
};

}  // namespace slamhip

#include "map_update_kernels.h"
#include "map_update_gather.h"

using namespace slamhip;

namespace {
constexpr int kNearR = 16, kNearSide = 2 * kNearR + 1, kNearMaxWords = 64;
struct MuScratch {
  size_t cap_records = 0, cap_beams = 0, temp_bytes = 0;
  unsigned *counts = nullptr, *offsets = nullptr, *keys = nullptr, *keys_sorted = nullptr;
  unsigned *order = nullptr, *order_sorted = nullptr;  // the beam of every record, before / after the sort
  double *beam_end = nullptr, *scan = nullptr;
  MuBeam *beam_info = nullptr;
  double *srt_prob = nullptr, *srt_qual = nullptr;  // sorted records
  int *occ = nullptr, *error_flag = nullptr;
  unsigned long long *n_updates = nullptr;  // one word: padding records of the update
  unsigned long long *h_status = nullptr;   // pinned: (error flag, padding records) of the last update
  void *temp = nullptr;
  // counting sort of a plain call's records (k_mu_rank): per-cell record counts of the key window (all zero between
  // updates), chain starts, the records as scattered
  size_t cap_bins = 0, scan_temp_bytes = 0;
  unsigned *bins = nullptr, *offs = nullptr;
  uint2 *srec = nullptr;  // (key, beam) of every record, as scattered into its cell's chain
  unsigned *h_offsets = nullptr;  // pinned: first record slot of every beam, computed by the host (x offset_slots)
  int offset_slots = 1;
  long long carried_updates = 0;   // cell updates of queued updates the ring collected by itself when it was full
  double *h_scan_stage = nullptr;  // pinned, kRing x 3 cap_beams (scans of up to 8192 points)
  int *h_occ_stage = nullptr;
  unsigned long long *near_bits = nullptr;  // [kNearSide^2][64] words: beams (up to 4096) visiting the cells next to the robot
  void *scan_temp = nullptr;
  // the gather form of a plain call (map_update_gather.h): per-beam closed forms, the irregular-cell bitmap of the key
  // window (all zero between updates), the workgroup counter of k_mu_cells, and what is known about the scanner's
  // beam directions -- analysed when they change (a scanner's angles do not change from scan to scan)
  int force_path = 0;  // SLAMHIP_OPT_K6_PATH of the context: 0 the fastest that applies, 1 counting sort, 2 radix sort
  MuLine *lines = nullptr;
  size_t cap_lines = 0, cap_irr_words = 0;
  unsigned *irr_bits = nullptr, *d_lut = nullptr;
  unsigned char *bad = nullptr;
  std::vector<double> geo_cos, geo_sin;
  bool geo_ok = false;
  double geo_a0 = 0.0;
  int geo_near_r = 8;
  // deferred completion (mu_set_deferred): updates are queued without waiting for them; their status words land in
  // a pinned ring and are summed up by mu_drain
  bool deferred = false;
  int pending = 0;
  unsigned long long *h_ring = nullptr;  // kRing x (error flag, padding records)
  unsigned ring_total[64] = {0};
  // ... written by the update kernels into slot `pending` of these device arrays and handed over by ONE
  // k_mu_finish_ring per drain (a finish kernel per update was 4.3 us of every particle's turn)
  int *d_ring_err = nullptr;
  unsigned long long *d_ring_pad = nullptr;
  // scan re-use (see slamhip_map_append_scan)
  bool reuse_ok = false;
  std::vector<double> raw_cos, raw_sin;  // slamhip_map_append_scan_raw: cos / sin(theta + a) of the call
  const double *last_range = nullptr, *last_cos = nullptr, *last_sin = nullptr;
  const int *last_occ = nullptr;
  const double *last_quality = nullptr;
  int last_n = -1;
};
// one scratch set per context, owned by it (contexts are independent: one caller thread each, and a
// process-wide registry would be shared state between those threads)
MuScratch &scratch_of(slamhip_ctx *ctx) {
  if (!ctx->mu_scratch) ctx->mu_scratch = new MuScratch;
  return *static_cast<MuScratch *>(ctx->mu_scratch);
}

}  // namespace

namespace slamhip {
constexpr int kRing = 64;
// internal: while enabled, a plain slamhip_map_append_scan on the zero-copy path returns as soon as its kernels are
// queued (n_updates_out = -1); the caller must not touch the scratch of this context from another stream and has
// to call mu_drain before it reads the map on the host or leaves.  The GMapping filter's shared-map loop uses it:
// the next particle's match is queued behind the update on the same stream, so nothing waits for the host.
bool mu_set_deferred(slamhip_ctx *ctx, bool on) {  // returns the previous setting
  MuScratch &sc = scratch_of(ctx);
  const bool was = sc.deferred;
  sc.deferred = on;
  return was;
}

// waits for the queued updates and adds up what they report
int mu_drain(slamhip_ctx *ctx, long long *n_updates, int *err) {
  MuScratch &sc = scratch_of(ctx);
  if (n_updates) *n_updates = sc.carried_updates;
  if (err) *err = 0;
  if (!sc.pending) {
    sc.carried_updates = 0;
    return SLAMHIP_OK;
  }
  unsigned seq = ++ctx->seq;
  if (seq == 0) seq = ++ctx->seq;
  hipLaunchKernelGGL(k_mu_finish_ring, dim3(1), dim3(64), 0, ctx->stream, sc.d_ring_err, sc.d_ring_pad, sc.pending,
                     sc.h_ring, ctx->h_done_flag, seq);
  SLAMHIP_CHECK(hipGetLastError());
  const int rc = score_wait(ctx, seq);
  if (rc) {
    sc.pending = 0;
    return rc;
  }
  long long nu = 0;
  int e = 0;
  for (int k = 0; k < sc.pending; ++k) {
    const unsigned long long fl = ((volatile unsigned long long *)sc.h_ring)[2 * k];
    const unsigned long long pad = ((volatile unsigned long long *)sc.h_ring)[2 * k + 1];
    if (fl) e = (int)fl;
    nu += (long long)sc.ring_total[k] - (long long)pad;
  }
  sc.pending = 0;
  nu += sc.carried_updates;
  sc.carried_updates = 0;
  if (n_updates) *n_updates = nu;
  if (err) *err = e;
  return SLAMHIP_OK;
}

// internal: while enabled, consecutive slamhip_map_append_scan calls that pass the very same host
// arrays upload them once (the caller guarantees their contents do not change in between)
void mu_allow_scan_reuse(slamhip_ctx *ctx, bool on) {
  MuScratch &sc = scratch_of(ctx);
  sc.reuse_ok = on;
  sc.last_n = -1;
}
}  // namespace slamhip

namespace {
int fail(const char *msg, int code = SLAMHIP_ERR_INVALID) {
  set_error(msg);
  return code;
}

// waits for the update queued on the context's stream and returns its status words
int mu_finish(slamhip_ctx *ctx, const int *d_error_flag, const unsigned long long *d_n_padding,
              unsigned long long *h_status, int *err, unsigned long long *n_padding) {
  if (ctx->low_latency) {
    unsigned seq = ++ctx->seq;
    if (seq == 0) seq = ++ctx->seq;
    hipLaunchKernelGGL(k_mu_finish, dim3(1), dim3(1), 0, ctx->stream, d_error_flag, d_n_padding, h_status,
                       ctx->h_done_flag, seq);
    SLAMHIP_CHECK(hipGetLastError());
    const int rc = score_wait(ctx, seq);
    if (rc) return rc;
    *err = (int)((volatile unsigned long long *)h_status)[0];
    *n_padding = ((volatile unsigned long long *)h_status)[1];
    return SLAMHIP_OK;
  }
  int e = 0;
  unsigned long long np = 0;
  SLAMHIP_CHECK(hipMemcpyAsync(&e, d_error_flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  SLAMHIP_CHECK(hipMemcpyAsync(&np, d_n_padding, sizeof(np), hipMemcpyDeviceToHost, ctx->stream));
  SLAMHIP_CHECK(hipStreamSynchronize(ctx->stream));
  *err = e;
  *n_padding = np;
  return SLAMHIP_OK;
}
}  // namespace

// The gather form needs the scan's beam directions in ascending order over less than a full turn (a laser scan's
// are), and a table from direction to beam index.  Both depend on the scanner only: analysed when cos_a / sin_a differ
// from the last call's (a 17 KB compare per update), not per scan.
int mu_scan_geometry(slamhip_ctx *ctx, MuScratch &sc, int n, const double *cos_a, const double *sin_a, bool *ok) {
  if ((int)sc.geo_cos.size() == n && std::memcmp(sc.geo_cos.data(), cos_a, sizeof(double) * n) == 0 &&
      std::memcmp(sc.geo_sin.data(), sin_a, sizeof(double) * n) == 0) {
    *ok = sc.geo_ok;
    return SLAMHIP_OK;
  }
  sc.geo_cos.assign(cos_a, cos_a + n);
  sc.geo_sin.assign(sin_a, sin_a + n);
  sc.geo_ok = false;
  *ok = false;
  std::vector<double> rel(n);
  double prev = 0.0, a0 = 0.0;
  for (int b = 0; b < n; ++b) {
    double ang = std::atan2(sin_a[b], cos_a[b]);
    if (!std::isfinite(ang)) return SLAMHIP_OK;
    if (b == 0) {
      a0 = ang;
      rel[0] = 0.0;
      prev = ang;
      continue;
    }
    while (ang < prev - 1e-12) ang += kTwoPi;  // unwrap: ascending
    if (!(ang >= prev) || ang - a0 >= kTwoPi - 0.05) return SLAMHIP_OK;  // not ascending within one turn
    rel[b] = ang - a0;
    prev = ang;
  }
  // the table over the PSEUDO-angle of a beam relative to beam 0 (mu_pseudo_angle: monotone in the angle)
  auto pseudo = [](double ang) {
    const double x = std::cos(ang), y = std::sin(ang);
    const double t = std::fabs(y) / (std::fabs(x) + std::fabs(y));
    return y >= 0.0 ? (x >= 0.0 ? t : 2.0 - t) : (x < 0.0 ? 2.0 + t : 4.0 - t);
  };
  std::vector<unsigned> lut(kGatherLut + 1);
  int b = 0;
  for (int m = 0; m <= kGatherLut; ++m) {
    const double edge = (double)m * (4.0 / kGatherLut);
    // (rel ascends within one turn, so its pseudo-angle ascends too -- except that angles a hair below a full turn
    // fold to ~4; they stay in front of `edge` only while rel itself is below the turn)
    while (b < n && (rel[b] < 1e-12 ? 0.0 : pseudo(rel[b])) < edge) ++b;
    lut[m] = (unsigned)b;
  }
  if (!sc.d_lut) SLAMHIP_CHECK(hipMalloc(&sc.d_lut, sizeof(unsigned) * (kGatherLut + 1)));
  SLAMHIP_CHECK(hipStreamSynchronize(ctx->stream));  // (queued updates may still read the old table)
  SLAMHIP_CHECK(hipMemcpy(sc.d_lut, lut.data(), sizeof(unsigned) * (kGatherLut + 1), hipMemcpyHostToDevice));
  sc.geo_a0 = a0;
  // cells closer to the robot than this take one wave each: beyond it a cell asks at most ~50 beams
  const double span = n > 1 ? rel[n - 1] : 1.0;
  const double density = n > 1 && span > 0 ? (double)(n - 1) / span : 1.0;  // beams per radian
  // (measured, 1080 beams over 270 degrees, k_mu_cells in us by radius: 8 cells 49, 12 cells 35, 18 cells 19, 26 and
  // 36 cells 19-20: a far cell's thread asks its candidates one after the other, a near cell's wave 64 at a time)
  const double half_window = std::min(1.5, 10.0 / density);
  sc.geo_near_r = std::max(2, std::min(40, (int)std::ceil(0.75 / std::sin(half_window))));
  sc.geo_ok = true;
  *ok = true;
  return SLAMHIP_OK;
}

extern "C" {

int slamhip_map_append_scan(slamhip_ctx *ctx, int map_id, const slamhip_scan_adder_cfg *cfg,
                            const double pose[3], int n, const double *range, const double *cos_a,
                            const double *sin_a, const int *is_occ, long long *n_updates_out) {
  return slamhip_map_append_scan_q(ctx, map_id, cfg, pose, n, range, cos_a, sin_a, is_occ, nullptr, n_updates_out);
}

// The reference's DEFAULT trig provider for the map update: append_scan sets the provider's base
// angle to the pose heading and moves every point with tp->cos / sin(angle) (grid_map_scan_adders.h:61-66,
// sensor_data.h:83-88) -- with RawTrigonometryProvider std::cos / std::sin(theta + a) per point
// (trigonometry_utils.h:17-35).  One call per scan: the HOST's libm evaluates them -- the reference's bits by
// definition, sin and cos one call each like the provider's two virtual functions (a fused sincos() is another
// build of the functions in glibc: csrc/libm_exact.h) -- and the update runs from the heading-free pose over the
// rotated directions: c = 1 c_b - 0 s_b = c_b, s = 0 c_b + 1 s_b = s_b exactly.
int slamhip_map_append_scan_raw(slamhip_ctx *ctx, int map_id, const slamhip_scan_adder_cfg *cfg, const double pose[3],
                                int n, const double *range, const double *angle, const int *is_occ, const double *quality,
                                long long *n_updates_out) {
  if (!ctx || !cfg || !pose || !range || !angle) return fail("null argument");
  if (n <= 0) return slamhip_map_append_scan_q(ctx, map_id, cfg, pose, n, range, angle, angle, is_occ, quality, n_updates_out);
  double (*volatile p_sin)(double) = ::sin;
  double (*volatile p_cos)(double) = ::cos;
  MuScratch &sc = scratch_of(ctx);
  sc.raw_cos.resize(n);
  sc.raw_sin.resize(n);
  for (int b = 0; b < n; ++b) {
    const double x = pose[2] + angle[b];  // _base_angle + angle_rad
    sc.raw_cos[b] = p_cos(x);
    sc.raw_sin[b] = p_sin(x);
  }
  sc.last_range = nullptr;  // (the arrays are rewritten in place per call: never "the scan of the call before")
  const double flat[3] = {pose[0], pose[1], 0.0};
  return slamhip_map_append_scan_q(ctx, map_id, cfg, flat, n, range, sc.raw_cos.data(), sc.raw_sin.data(), is_occ, quality,
                                   n_updates_out);
}

int slamhip_map_append_scan_q(slamhip_ctx *ctx, int map_id, const slamhip_scan_adder_cfg *cfg,
                              const double pose[3], int n, const double *range, const double *cos_a,
                              const double *sin_a, const int *is_occ, const double *quality,
                              long long *n_updates_out) {
  if (!ctx || !cfg || !pose || !range || !cos_a || !sin_a) return fail("null argument");
  if (map_id < 0 || map_id >= (int)ctx->maps.size() || !ctx->maps[map_id].bound) return fail("unknown map id");
  if (n <= 0) {
    if (n_updates_out) *n_updates_out = 0;
    return SLAMHIP_OK;
  }
  if (n > 8192 * 1024) return fail("too many scan points");
  DeviceMap &m = ctx->maps[map_id];
  const int rule = cfg->rule;
  if (rule < SLAMHIP_RULE_LAST || rule > SLAMHIP_RULE_GMAPPING) return fail("unknown cell update rule");
  if (cfg->occupancy_estimator != 0 && cfg->occupancy_estimator != 1) return fail("unknown occupancy estimator");
  const bool ok_model = (rule == SLAMHIP_RULE_TBM && m.cell_model == SLAMHIP_CELL_TBM) ||
                        (rule == SLAMHIP_RULE_GMAPPING && m.cell_model == SLAMHIP_CELL_GMAPPING) ||
                        (rule <= SLAMHIP_RULE_MEAN && m.cell_model == SLAMHIP_CELL_OCC);
  if (!ok_model) return fail("cell update rule does not fit the map's payload model");
  SLAMHIP_CHECK(hipSetDevice(ctx->device));
  const int aux_stride = rule == SLAMHIP_RULE_MEAN ? 1 : (rule == SLAMHIP_RULE_GMAPPING ? 2 : 0);
  if (aux_stride && (m.aux_stride != aux_stride || !m.d_aux)) {
    if (m.d_aux) hipFree(m.d_aux);
    m.d_aux = nullptr;
    const size_t bytes = (size_t)m.pitch * m.height * aux_stride * sizeof(double);
    SLAMHIP_CHECK(hipMalloc(&m.d_aux, bytes));
    SLAMHIP_CHECK(hipMemsetAsync(m.d_aux, 0, bytes, ctx->stream));
    m.aux_stride = aux_stride;
  }
  MuScratch &sc = scratch_of(ctx);
  if ((size_t)n > sc.cap_beams) {
    SLAMHIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (void *p : {(void *)sc.counts, (void *)sc.offsets, (void *)sc.beam_end, (void *)sc.scan, (void *)sc.occ,
                    (void *)sc.beam_info})
      if (p) hipFree(p);
    size_t cap = 2048;
    while (cap < (size_t)n) cap *= 2;
    SLAMHIP_CHECK(hipMalloc(&sc.counts, sizeof(unsigned) * cap));
    SLAMHIP_CHECK(hipMalloc(&sc.offsets, sizeof(unsigned) * (cap + 1)));
    SLAMHIP_CHECK(hipMalloc(&sc.beam_end, sizeof(double) * 4 * cap));  // end point + (1/dx, 1/dy)
    SLAMHIP_CHECK(hipMalloc(&sc.beam_info, sizeof(MuBeam) * cap));
    SLAMHIP_CHECK(hipMalloc(&sc.scan, sizeof(double) * 4 * cap));  // range | cos | sin | per-point quality
    SLAMHIP_CHECK(hipMalloc(&sc.occ, sizeof(int) * cap));
    if (!sc.error_flag) {
      SLAMHIP_CHECK(hipMalloc(&sc.error_flag, sizeof(int)));
      SLAMHIP_CHECK(hipMemsetAsync(sc.error_flag, 0, sizeof(int), ctx->stream));
    }
    // (queued updates: one slot per update in flight -- the kernels of an update read their slot when they run,
    // not when they are queued; scans too large for 64 slots are awaited)
    if (sc.h_offsets) hipHostFree(sc.h_offsets);
    sc.offset_slots = cap <= 65536 ? kRing : 1;
    SLAMHIP_CHECK(hipHostMalloc(&sc.h_offsets, sizeof(unsigned) * (cap + 1) * sc.offset_slots, hipHostMallocDefault));
    // ... and of the scan itself (lidar-sized scans): range | cos | sin packed in pinned memory and sent with one
    // asynchronous copy instead of three staged ones, the point flags with a second
    if (sc.h_scan_stage) hipHostFree(sc.h_scan_stage);
    if (sc.h_occ_stage) hipHostFree(sc.h_occ_stage);
    sc.h_scan_stage = nullptr;
    sc.h_occ_stage = nullptr;
    if (cap <= 8192) {
      SLAMHIP_CHECK(hipHostMalloc(&sc.h_scan_stage, sizeof(double) * 4 * cap * kRing, hipHostMallocDefault));
      SLAMHIP_CHECK(hipHostMalloc(&sc.h_occ_stage, sizeof(int) * cap * kRing, hipHostMallocDefault));
    }
    if (!sc.near_bits)
      SLAMHIP_CHECK(hipMalloc(&sc.near_bits, sizeof(unsigned long long) * kNearSide * kNearSide * kNearMaxWords));
    if (!sc.n_updates) {
      SLAMHIP_CHECK(hipMalloc(&sc.n_updates, sizeof(unsigned long long)));
      SLAMHIP_CHECK(hipMemsetAsync(sc.n_updates, 0, sizeof(unsigned long long), ctx->stream));
    }
    if (!sc.h_status) SLAMHIP_CHECK(hipHostMalloc(&sc.h_status, 2 * sizeof(unsigned long long), hipHostMallocDefault));
    if (!sc.h_ring) SLAMHIP_CHECK(hipHostMalloc(&sc.h_ring, 2 * kRing * sizeof(unsigned long long), hipHostMallocDefault));
    if (!sc.d_ring_err) {
      SLAMHIP_CHECK(hipMalloc(&sc.d_ring_err, kRing * sizeof(int)));
      SLAMHIP_CHECK(hipMalloc(&sc.d_ring_pad, kRing * sizeof(unsigned long long)));
      SLAMHIP_CHECK(hipMemsetAsync(sc.d_ring_err, 0, kRing * sizeof(int), ctx->stream));
      SLAMHIP_CHECK(hipMemsetAsync(sc.d_ring_pad, 0, kRing * sizeof(unsigned long long), ctx->stream));
    }
    sc.cap_beams = cap;
  }
  const size_t cb = sc.cap_beams;
  // the GMapping filter appends the SAME raw scan once per particle: it brackets its loop with
  // mu_allow_scan_reuse(), and identical host arrays are then uploaded only once
  const bool reuse = sc.reuse_ok && sc.last_range == range && sc.last_cos == cos_a && sc.last_sin == sin_a &&
                     sc.last_occ == is_occ && sc.last_n == n && !quality && !sc.last_quality;
  const bool deferred = sc.deferred && ctx->low_latency && sc.offset_slots == kRing;
  if (deferred && sc.pending == kRing) {  // every slot of the ring is taken: collect first
    long long dn = 0;
    int de = 0;
    const int drc = mu_drain(ctx, &dn, &de);
    if (drc) return drc;
    sc.carried_updates += dn;
    if (de) return fail("a deferred map update reported an error (beam outside the window?)", SLAMHIP_ERR_STATE);
  }
  if (!reuse && sc.h_scan_stage) {
    // (slot `pending` while updates are queued -- its last user has been drained; slot 0 otherwise: awaited)
    const size_t slot = deferred ? (size_t)sc.pending : 0;
    double *st = sc.h_scan_stage + slot * 4 * cb;
    std::memcpy(st, range, sizeof(double) * n);
    std::memcpy(st + cb, cos_a, sizeof(double) * n);
    std::memcpy(st + 2 * cb, sin_a, sizeof(double) * n);
    if (quality) std::memcpy(st + 3 * cb, quality, sizeof(double) * n);
    SLAMHIP_CHECK(hipMemcpyAsync(sc.scan, st, sizeof(double) * ((quality ? 3 : 2) * cb + n), hipMemcpyHostToDevice, ctx->stream));
    if (is_occ) {
      int *so = sc.h_occ_stage + slot * cb;
      std::memcpy(so, is_occ, sizeof(int) * n);
      SLAMHIP_CHECK(hipMemcpyAsync(sc.occ, so, sizeof(int) * n, hipMemcpyHostToDevice, ctx->stream));
    }
  } else if (!reuse) {
    SLAMHIP_CHECK(hipMemcpyAsync(sc.scan, range, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
    SLAMHIP_CHECK(hipMemcpyAsync(sc.scan + cb, cos_a, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
    SLAMHIP_CHECK(hipMemcpyAsync(sc.scan + 2 * cb, sin_a, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
    if (quality) SLAMHIP_CHECK(hipMemcpyAsync(sc.scan + 3 * cb, quality, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
    if (is_occ) SLAMHIP_CHECK(hipMemcpyAsync(sc.occ, is_occ, sizeof(int) * n, hipMemcpyHostToDevice, ctx->stream));
  }
  if (!reuse) {
    sc.last_range = range;
    sc.last_cos = cos_a;
    sc.last_sin = sin_a;
    sc.last_occ = is_occ;
    sc.last_n = n;
    sc.last_quality = quality;
  }

  MuArgs a;
  std::memset(&a, 0, sizeof(a));
  a.n_jobs = 1;
  auto fill_map = [&]() {
    a.payload = m.d_payload;
    a.aux = aux_stride ? m.d_aux : nullptr;
    a.width = m.width;
    a.height = m.height;
    a.pitch = m.pitch;
    a.origin_x = m.origin_x;
    a.origin_y = m.origin_y;
    // (a window that grew a moment ago has no masks: its next GMapping scorer call derives them)
    a.nbr_on = (m.nbr_ok && m.cell_model == SLAMHIP_CELL_GMAPPING) ? 1 : 0;
    a.nbr_th = m.nbr_th;
    a.prob = (m.prob_ok && m.cell_model == SLAMHIP_CELL_TBM) ? m.d_prob : nullptr;  // (gone after a re-bind, like the masks)
  };
  fill_map();
  a.cell_dbl = cell_doubles(m.cell_model);
  a.aux_stride = aux_stride;
  a.scale = m.scale;
  a.range = sc.scan;
  a.cos_a = sc.scan + cb;
  a.sin_a = sc.scan + 2 * cb;
  a.is_occ = is_occ ? sc.occ : nullptr;
  // per-point observation quality (ObservationMappingQualityEstimator::quality, grid_map_scan_adders.h:17-43):
  // a cell update's quality is scan_quality x the value of ITS beam; null = IdleOMQE
  a.beam_quality = quality ? sc.scan + 3 * cb : nullptr;
  a.n = n;
  a.px = pose[0];
  a.py = pose[1];
  ::sincos(pose[2], &a.sn, &a.cs);  // set_base_angle(pose.theta), grid_map_scan_adders.h:61
  a.rule = rule;
  a.est_kind = cfg->occupancy_estimator;
  a.shift_amount = cfg->area_shift_amount > 0 ? cfg->area_shift_amount : 0.01 * m.scale;
  a.quality = cfg->scan_quality * 1.0;  // IdleOMQE
  a.base_occ_prob = cfg->base_occupied_prob;
  a.base_occ_qual = cfg->base_occupied_qual;
  a.base_empty_prob = cfg->base_empty_prob;
  a.base_empty_qual = cfg->base_empty_qual;
  a.blur = cfg->blur;
  a.max_range_sq = cfg->max_range * cfg->max_range;
  a.counts = sc.counts;
  a.offsets = sc.offsets;
  a.beam_end = sc.beam_end;
  a.beam_inv = sc.beam_end + 2 * sc.cap_beams;
  a.beam_info = sc.beam_info;
  a.error_flag = sc.error_flag;
  a.n_padding = sc.n_updates;
  if (deferred) {  // queued, not awaited: the status words go to slot `pending` of the ring (mu_drain)
    a.error_flag = sc.d_ring_err + sc.pending;
    a.n_padding = sc.d_ring_pad + sc.pending;
  }

  const dim3 bgrid((n + 255) / 256);
  ProfilePairGuard prof;  // slamhip_profile_read_map_update: the whole pipeline (closed on every exit)
  {
    const int prc = prof.open(ctx, ctx->stream, 1);
    if (prc) return prc;
  }
  unsigned *const h_off = sc.h_offsets + (deferred ? (size_t)sc.pending * (sc.cap_beams + 1) : 0);
  a.host_offsets = h_off;
  const int near_words = (n + 63) / 64;
  if (near_words <= kNearMaxWords) {
    a.near_bits = sc.near_bits;
    a.near_r = kNearR;
    a.near_words = near_words;
  }
  // the record count is needed on the host to size the buffers: the same IEEE operations as
  // k_mu_count (no contraction on either side) give the same bounds without a device round trip
  unsigned total = 0;
  int bb_lo_x = 0, bb_lo_y = 0, bb_hi_x = 0, bb_hi_y = 0;  // external cells the update can touch: robot cell .. end cells
  auto host_pass = [&]() -> int {
    const int rcx = (int)std::floor(a.px / a.scale), rcy = (int)std::floor(a.py / a.scale);
    bb_lo_x = bb_hi_x = rcx;
    bb_lo_y = bb_hi_y = rcy;
    for (int b = 0; b < n; ++b) {
      h_off[b] = total;
      const double c = a.cs * cos_a[b] - a.sn * sin_a[b];
      const double s = a.sn * cos_a[b] + a.cs * sin_a[b];
      const double wx = a.px + range[b] * c, wy = a.py + range[b] * s;
      const double ddx = wx - a.px, ddy = wy - a.py;
      if (a.max_range_sq < ddx * ddx + ddy * ddy) continue;
      // a beam that is ray-traced needs a cell: the reference asserts here (regular_squares_grid.h:42), and
      // int(floor(inf or NaN)) is not the same number on the host and on the device -- the record buffers
      // are sized from THIS loop and filled by the device's
      if (!(std::fabs(wx / a.scale) < 1073741824.0) || !(std::fabs(wy / a.scale) < 1073741824.0))
        return fail("a scan point that is not range-gated has a non-finite or absurdly far end point "
                    "(no-return beams need a finite range or slam/mapping/max_range)");
      const int ocx = (int)std::floor(wx / a.scale), ocy = (int)std::floor(wy / a.scale);
      total += (unsigned)(std::abs(ocx - rcx) + std::abs(ocy - rcy) + 1);
      bb_lo_x = std::min(bb_lo_x, ocx);
      bb_hi_x = std::max(bb_hi_x, ocx);
      bb_lo_y = std::min(bb_lo_y, ocy);
      bb_hi_y = std::max(bb_hi_y, ocy);
    }
    return SLAMHIP_OK;
  };
  auto launch_count = [&]() {
    if (a.est_kind == 1) hipLaunchKernelGGL(k_mu_count<1>, bgrid, dim3(256), 0, ctx->stream, a);
    else hipLaunchKernelGGL(k_mu_count<0>, bgrid, dim3(256), 0, ctx->stream, a);
  };
  if (!m.auto_grow) {
    const int hrc = host_pass();
    if (hrc) return hrc;
  } else {
    // an unbounded map (slamhip_map_set_auto_grow): the window first grows to hold every cell of the update, as
    // UnboundedPlainGridMap::update does cell by cell (plain_grid_map.h:62-67,133-173).  Any superset of the
    // reference's window holds the same cells; this one adds a fifth of the side (the reference's
    // Expansion_Rate) or what the scan needs, whichever is more, on the sides that were left.
    const int hrc = host_pass();
    if (hrc) return hrc;
    const long long lo_x = (long long)bb_lo_x + m.origin_x, hi_x = (long long)bb_hi_x + m.origin_x;
    const long long lo_y = (long long)bb_lo_y + m.origin_y, hi_y = (long long)bb_hi_y + m.origin_y;
    if (total > 0 && (lo_x < 0 || lo_y < 0 || hi_x >= m.width || hi_y >= m.height)) {
      auto side = [](long long need, int dim) { return need > 0 ? std::max(need, (long long)dim / 5 + 1) : 0ll; };
      const long long px = side(-lo_x, m.width), ax = side(hi_x - (m.width - 1), m.width);
      const long long py = side(-lo_y, m.height), ay = side(hi_y - (m.height - 1), m.height);
      const long long nw = px + m.width + ax, nh = py + m.height + ay;
      if (nw * nh > (1ll << 31))
        return fail("an unbounded map would grow beyond 2^31 cells: a scan point far outside the site?", SLAMHIP_ERR_STATE);
      const long grown = m.grown + 1;
      const int model = m.cell_model;
      const double sc_ = m.scale;
      double unk[4];
      for (int k = 0; k < 4; ++k) unk[k] = m.unknown[k];
      const int rc = slamhip_map_bind(ctx, map_id, model, (int)nw, (int)nh, m.origin_x + (int)px, m.origin_y + (int)py, sc_, unk);
      if (rc) return rc;
      ctx->maps[map_id].grown = grown;
      fill_map();
    }
  }
  if (total == 0) {
    if (n_updates_out) *n_updates_out = 0;
    return SLAMHIP_OK;
  }
  if (total > sc.cap_records) {
    SLAMHIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (void *p : {(void *)sc.keys, (void *)sc.keys_sorted, (void *)sc.order, (void *)sc.order_sorted,
                    (void *)sc.srt_prob, (void *)sc.srt_qual, sc.temp, (void *)sc.srec})
      if (p) hipFree(p);
    sc.srec = nullptr;
    size_t cap = 1 << 16;
    while (cap < total) cap *= 2;
    SLAMHIP_CHECK(hipMalloc(&sc.keys, sizeof(unsigned) * cap));
    SLAMHIP_CHECK(hipMalloc(&sc.keys_sorted, sizeof(unsigned) * cap));
    SLAMHIP_CHECK(hipMalloc(&sc.order, sizeof(unsigned) * cap));
    SLAMHIP_CHECK(hipMalloc(&sc.order_sorted, sizeof(unsigned) * cap));
    SLAMHIP_CHECK(hipMalloc(&sc.srt_prob, sizeof(double) * cap));
    SLAMHIP_CHECK(hipMalloc(&sc.srt_qual, sizeof(double) * cap));
    sc.temp_bytes = 0;
    SLAMHIP_CHECK(rocprim::radix_sort_pairs(nullptr, sc.temp_bytes, sc.keys, sc.keys_sorted, sc.order,
                                            sc.order_sorted, cap, 0, 32, ctx->stream));
    SLAMHIP_CHECK(hipMalloc(&sc.temp, sc.temp_bytes));
    sc.cap_records = cap;
  }
  a.keys = sc.keys;
  a.keys_cap = (unsigned long long)sc.cap_records;
  // the key window: the cells between the robot's and the beams' end cells, clipped to the map (a walk is monotone
  // between its two ends; cells outside the map become padding).  Small enough -- a few hundred thousand cells for
  // a laser's reach -- the records are counting-sorted over it; otherwise keys are cells of the whole map and
  // rocprim sorts them (SLAMHIP_OPT_K6_PATH forces either path: the parity tests run all three).
  sc.force_path = ctx->k6_path;
  const bool force_radix = sc.force_path == 2;
  const long long wx0 = std::max(0ll, (long long)bb_lo_x + m.origin_x), wy0 = std::max(0ll, (long long)bb_lo_y + m.origin_y);
  const long long wx1 = std::min((long long)m.width - 1, (long long)bb_hi_x + m.origin_x);
  const long long wy1 = std::min((long long)m.height - 1, (long long)bb_hi_y + m.origin_y);
  const long long n_bins = (wx1 >= wx0 && wy1 >= wy0) ? (wx1 - wx0 + 1) * (wy1 - wy0 + 1) : 0;
  const bool counting = !force_radix && n_bins > 0 && n_bins <= (1ll << 23) && near_words <= kNearMaxWords;
  if (counting && (size_t)n_bins + 1 > sc.cap_bins) {
    SLAMHIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (void *p : {(void *)sc.bins, (void *)sc.offs, sc.scan_temp})
      if (p) hipFree(p);
    sc.bins = sc.offs = nullptr;
    sc.scan_temp = nullptr;
    size_t cap = 1 << 18;
    while (cap < (size_t)n_bins + 1) cap *= 2;
    SLAMHIP_CHECK(hipMalloc(&sc.bins, sizeof(unsigned) * cap));
    SLAMHIP_CHECK(hipMemsetAsync(sc.bins, 0, sizeof(unsigned) * cap, ctx->stream));
    SLAMHIP_CHECK(hipMalloc(&sc.offs, sizeof(unsigned) * cap));
    sc.scan_temp_bytes = 0;
    SLAMHIP_CHECK(rocprim::exclusive_scan(nullptr, sc.scan_temp_bytes, sc.bins, sc.offs, 0u, cap, rocprim::plus<unsigned>(),
                                          ctx->stream));
    SLAMHIP_CHECK(hipMalloc(&sc.scan_temp, sc.scan_temp_bytes));
    sc.cap_bins = cap;
  }
  // The GATHER form (map_update_gather.h, the default wherever it applies): two kernels, no records.
  // SLAMHIP_OPT_K6_PATH = 1 / 2 keep the record pipelines (the parity tests run all three).
  const bool force_counting = sc.force_path == 1;
  bool gather = counting && !force_counting && ctx->low_latency && n <= 4096 && n_bins <= (1ll << 22);
  if (gather) {
    const int grc = mu_scan_geometry(ctx, sc, n, cos_a, sin_a, &gather);
    if (grc) return grc;
  }
  if (gather) {
    if ((size_t)n > sc.cap_lines) {
      SLAMHIP_CHECK(hipStreamSynchronize(ctx->stream));
      if (sc.lines) hipFree(sc.lines);
      sc.lines = nullptr;
      SLAMHIP_CHECK(hipMalloc(&sc.lines, sizeof(MuLine) * sc.cap_beams));
      if (sc.bad) hipFree(sc.bad);
      sc.bad = nullptr;
      SLAMHIP_CHECK(hipMalloc(&sc.bad, sc.cap_beams + 8));
      SLAMHIP_CHECK(hipMemsetAsync(sc.bad, 0, sc.cap_beams + 8, ctx->stream));
      sc.cap_lines = sc.cap_beams;
    }
    const size_t words = (size_t)n_bins + 1;  // (a marker word per window cell)
    if (words > sc.cap_irr_words) {
      SLAMHIP_CHECK(hipStreamSynchronize(ctx->stream));
      if (sc.irr_bits) hipFree(sc.irr_bits);
      sc.irr_bits = nullptr;
      size_t cap = 1 << 14;
      while (cap < words) cap *= 2;
      SLAMHIP_CHECK(hipMalloc(&sc.irr_bits, sizeof(unsigned) * cap));
      SLAMHIP_CHECK(hipMemsetAsync(sc.irr_bits, 0, sizeof(unsigned) * cap, ctx->stream));
      sc.cap_irr_words = cap;
    }
    a.key_x0 = (int)wx0;
    a.key_y0 = (int)wy0;
    a.key_w = (int)(wx1 - wx0 + 1);
    a.key_h = (int)(wy1 - wy0 + 1);
    a.n_bins = (unsigned)n_bins;
    a.bins = nullptr;
    a.near_bits = nullptr;
    a.near_r = sc.geo_near_r;
    a.robot_ix = (int)std::floor(a.px / a.scale) + m.origin_x;
    a.robot_iy = (int)std::floor(a.py / a.scale) + m.origin_y;
    a.lines = sc.lines;
    a.lut = sc.d_lut;
    a.lut_bins = kGatherLut;
    ::sincos(pose[2] + sc.geo_a0, &a.rot_s, &a.rot_c);
    a.irr_bits = sc.irr_bits;
    a.bad = sc.bad;
    if (a.est_kind == 1) hipLaunchKernelGGL(k_mu_lines<1>, dim3(n), dim3(256), 0, ctx->stream, a);
    else hipLaunchKernelGGL(k_mu_lines<0>, dim3(n), dim3(256), 0, ctx->stream, a);
    const unsigned far_blocks = (unsigned)(((a.key_w + 15) / 16) * ((a.key_h + 15) / 16));
    const unsigned side = 2u * (unsigned)a.near_r + 1u, near_blocks = (side * side + 3u) / 4u;
    const dim3 cgrid(far_blocks + near_blocks), cblock(256);
#define SLAMHIP_MU_CELLS(R)                                                                                          \
  case R:                                                                                                            \
    if (a.est_kind == 1)                                                                                             \
      hipLaunchKernelGGL((k_mu_cells<R, 1>), cgrid, cblock, 0, ctx->stream, a, far_blocks);                           \
    else                                                                                                             \
      hipLaunchKernelGGL((k_mu_cells<R, 0>), cgrid, cblock, 0, ctx->stream, a, far_blocks);                           \
    break;
    switch (a.rule) {
      SLAMHIP_MU_CELLS(0)
      SLAMHIP_MU_CELLS(1)
      SLAMHIP_MU_CELLS(2)
      SLAMHIP_MU_CELLS(3)
      default:
        SLAMHIP_MU_CELLS(4)
    }
#undef SLAMHIP_MU_CELLS
    SLAMHIP_CHECK(hipGetLastError());
    if (prof.on()) {
      const int prc = prof.close();
      if (prc) return prc;
      ctx->prof_k6_calls += 1;
      ctx->prof_k6_records += total;
    }
    if (deferred) {
      sc.ring_total[sc.pending] = total;
      ++sc.pending;
      if (n_updates_out) *n_updates_out = -1;
      return SLAMHIP_OK;
    }
    // (a same-address counter that let the last workgroup hand the status over cost 42 us with 700 workgroups:
    // the one-thread kernel behind the update is 4)
    int gerr = 0;
    unsigned long long gpad = 0;
    const int wrc = mu_finish(ctx, sc.error_flag, sc.n_updates, sc.h_status, &gerr, &gpad);
    if (wrc) return wrc;
    if (n_updates_out) *n_updates_out = (long long)((unsigned long long)total - gpad);
    if (gerr == 2) return fail("internal: the device counted more cell updates than the host sized the buffers for", SLAMHIP_ERR_STATE);
    if (gerr)
      return fail("a beam leaves the bound map window: grow the map (slamhip_map_bind) before updating; "
                  "cells inside the window were updated", SLAMHIP_ERR_STATE);
    return SLAMHIP_OK;
  }
  if (counting && !sc.srec) SLAMHIP_CHECK(hipMalloc(&sc.srec, sizeof(uint2) * sc.cap_records));
  if (counting) {
    a.key_x0 = (int)wx0;
    a.key_y0 = (int)wy0;
    a.key_w = (int)(wx1 - wx0 + 1);
    a.bins = sc.bins;
    a.n_bins = (unsigned)n_bins;
    a.near_bits = sc.near_bits;
    a.near_r = kNearR;
    a.near_words = near_words;
    a.robot_ix = (int)std::floor(a.px / a.scale) + m.origin_x;
    a.robot_iy = (int)std::floor(a.py / a.scale) + m.origin_y;
  } else {
    a.near_bits = nullptr;
    a.key_x0 = a.key_y0 = 0;
    a.key_w = m.pitch;
  }
  // counting-sorted updates on the zero-copy path need no k_mu_count: k_mu_emit does its work, k_mu_finish of the
  // update before cleared the status word
  const bool fused = counting && ctx->low_latency;
  if (!fused) launch_count();
  if (!fused) hipLaunchKernelGGL((k_mu_emit<unsigned, -1>), dim3((n + 3) / 4), dim3(256), 0, ctx->stream, a, sc.order);
  else if (a.est_kind == 1) hipLaunchKernelGGL((k_mu_emit<unsigned, 1>), dim3((n + 3) / 4), dim3(256), 0, ctx->stream, a, sc.order);
  else hipLaunchKernelGGL((k_mu_emit<unsigned, 0>), dim3((n + 3) / 4), dim3(256), 0, ctx->stream, a, sc.order);
  if (counting) {
    hipLaunchKernelGGL(k_mu_near_bits, dim3(near_words, kNearSide), dim3(64), 0, ctx->stream, a);
    size_t tb = sc.scan_temp_bytes;
    SLAMHIP_CHECK(rocprim::exclusive_scan(sc.scan_temp, tb, sc.bins, sc.offs, 0u, (size_t)n_bins + 1, rocprim::plus<unsigned>(),
                                          ctx->stream));
    const dim3 rgrid((total + 255) / 256);
    hipLaunchKernelGGL(k_mu_scatter, rgrid, dim3(256), 0, ctx->stream, a, (const unsigned *)sc.keys, (const unsigned *)sc.order,
                       total, (const unsigned *)sc.offs, sc.srec);
    if (a.est_kind == 1)
      hipLaunchKernelGGL(k_mu_rank<1>, rgrid, dim3(256), 0, ctx->stream, a, (const uint2 *)sc.srec,
                         (const unsigned *)sc.offs, total, sc.keys_sorted, sc.order_sorted, sc.srt_prob, sc.srt_qual);
    else
      hipLaunchKernelGGL(k_mu_rank<0>, rgrid, dim3(256), 0, ctx->stream, a, (const uint2 *)sc.srec,
                         (const unsigned *)sc.offs, total, sc.keys_sorted, sc.order_sorted, sc.srt_prob, sc.srt_qual);
  } else {
    size_t tb = sc.temp_bytes;
    // sort only the bits a cell key can occupy; the invalid key (all ones) still sorts last because
    // every valid key is < 2^nbits - 1
    unsigned nbits = 1;
    while (nbits < 32 && ((1ull << nbits) - 1) <= (unsigned long long)m.pitch * m.height) ++nbits;
    SLAMHIP_CHECK(rocprim::radix_sort_pairs(sc.temp, tb, sc.keys, sc.keys_sorted, sc.order, sc.order_sorted,
                                            total, 0, nbits, ctx->stream));
    if (a.est_kind == 1)
      hipLaunchKernelGGL((k_mu_gather<unsigned, 1>), dim3((total + 255) / 256), dim3(256), 0, ctx->stream, a,
                         (const unsigned *)sc.keys_sorted, (const unsigned *)sc.order_sorted, total, sc.srt_prob,
                         sc.srt_qual);
    else
      hipLaunchKernelGGL((k_mu_gather<unsigned, 0>), dim3((total + 255) / 256), dim3(256), 0, ctx->stream, a,
                         (const unsigned *)sc.keys_sorted, (const unsigned *)sc.order_sorted, total, sc.srt_prob,
                         sc.srt_qual);
  }
  a.rec_prob = sc.srt_prob;
  a.rec_qual = sc.srt_qual;
  a.rec_beam = sc.order_sorted;
  mu_launch_apply<unsigned>(a, (const unsigned *)sc.keys_sorted, total, ctx->stream);
  SLAMHIP_CHECK(hipGetLastError());
  if (prof.on()) {
    const int prc = prof.close();
    if (prc) return prc;
    ctx->prof_k6_calls += 1;
    ctx->prof_k6_records += total;
  }
  if (deferred) {
    sc.ring_total[sc.pending] = total;
    ++sc.pending;
    if (n_updates_out) *n_updates_out = -1;
    return SLAMHIP_OK;
  }
  int err = 0;
  unsigned long long nu = 0;  // padding records
  {
    const int rc = mu_finish(ctx, sc.error_flag, sc.n_updates, sc.h_status, &err, &nu);
    if (rc) return rc;
  }
  nu = (unsigned long long)total - nu;
  if (n_updates_out) *n_updates_out = (long long)nu;
  if (err == 2) return fail("internal: the device counted more cell updates than the host sized the buffers for", SLAMHIP_ERR_STATE);
  if (err)
    return fail("a beam leaves the bound map window: grow the map (slamhip_map_bind) before updating; "
                "cells inside the window were updated", SLAMHIP_ERR_STATE);
  return SLAMHIP_OK;
}

int slamhip_map_download_aux(slamhip_ctx *ctx, int map_id, int x0, int y0, int w, int h, double *out) {
  if (!ctx || !out) return fail("null argument");
  if (map_id < 0 || map_id >= (int)ctx->maps.size() || !ctx->maps[map_id].bound) return fail("unknown map id");
  DeviceMap &m = ctx->maps[map_id];
  if (!m.d_aux || !m.aux_stride) return fail("this map holds no update counters", SLAMHIP_ERR_STATE);
  if (w <= 0 || h <= 0 || x0 < 0 || y0 < 0 || x0 + w > m.width || y0 + h > m.height)
    return fail("window outside the bound map");
  const size_t cb = m.aux_stride * sizeof(double);
  SLAMHIP_CHECK(hipMemcpy2DAsync(out, w * cb, m.d_aux + ((size_t)y0 * m.pitch + x0) * m.aux_stride, m.pitch * cb,
                                 w * cb, h, hipMemcpyDeviceToHost, ctx->stream));
  SLAMHIP_CHECK(hipStreamSynchronize(ctx->stream));
  return SLAMHIP_OK;
}

}  // extern "C"

// ---- batch: the same scan appended from many poses, each into its own copy-on-write map ----------
#include "tile_pool.h"

namespace {
struct MuBatchScratch {
  size_t cap_beams = 0, cap_records = 0, cap_jobs = 0, cap_scan = 0, temp_bytes = 0;
  unsigned *counts = nullptr, *offsets = nullptr, *order = nullptr, *order_sorted = nullptr;
  unsigned long long *keys = nullptr, *keys_sorted = nullptr;
  double *beam_end = nullptr, *scan = nullptr;
  MuBeam *beam_info = nullptr;
  double *srt_prob = nullptr;
  int *occ = nullptr, *error_flag = nullptr;
  MuJob *d_jobs = nullptr;
  int *d_bbox = nullptr;
  unsigned long long *n_updates = nullptr, *d_total = nullptr, *h_status = nullptr;
  void *temp = nullptr, *scan_temp = nullptr;
  size_t scan_temp_bytes = 0;
  // free-space fast path (mu_batch_fast_tail): the marked cells, per-beam count / place of the records left to sort
  unsigned *special = nullptr, *slow_cnt = nullptr, *slow_off = nullptr;
  size_t special_words = 0, cap_slow = 0;
};
MuBatchScratch &bscratch_of(slamhip_ctx *ctx) {
  if (!ctx->mu_bscratch) ctx->mu_bscratch = new MuBatchScratch;
  return *static_cast<MuBatchScratch *>(ctx->mu_bscratch);
}

template <typename T>
hipError_t regrow(T *&p, size_t count) {
  if (p) hipFree(p);
  p = nullptr;
  return hipMalloc(&p, sizeof(T) * count);
}

// rocprim's onesweep for the batch sort.  4-byte keys: 8 bits per pass, 1024 x 8 items per block, `match`
// ranking -- 611 us against 750 us for rocprim's gfx950 default on 21.6 M random (key, value) pairs
// (tools/sort_probe.hip); other shapes measured there were slower or exceed the LDS.
template <typename Key>
struct BatchSortConfigOf {
  using type = rocprim::default_config;
};
template <>
struct BatchSortConfigOf<unsigned> {
  using type = rocprim::radix_sort_config<
      rocprim::default_config, rocprim::default_config,
      rocprim::radix_sort_onesweep_config<rocprim::kernel_config<1024, 8>, rocprim::kernel_config<1024, 8>, 8,
                                          rocprim::block_radix_rank_algorithm::match>>;
};
template <typename Key>
using BatchSortConfig = typename BatchSortConfigOf<Key>::type;

// walk, sort, evaluate, apply -- with the key width the batch's (job, window cell) pairs need
template <typename Key>
int mu_batch_tail(const MuArgs &a, MuBatchScratch &sc, unsigned total, size_t beams, unsigned end_bit, hipStream_t st) {
  Key *keys = (Key *)sc.keys, *keys_sorted = (Key *)sc.keys_sorted;
  const dim3 bgrid((unsigned)((beams + 255) / 256)), rgrid((total + 255) / 256);
  hipLaunchKernelGGL((k_mu_emit<Key, -1>), dim3((unsigned)((beams + 3) / 4)), dim3(256), 0, st, a, sc.order);
  size_t tb = sc.temp_bytes;
  SLAMHIP_CHECK(rocprim::radix_sort_pairs<BatchSortConfig<Key>>(sc.temp, tb, keys, keys_sorted, sc.order,
                                                                sc.order_sorted, total, 0,
                                                                std::min(end_bit, (unsigned)(8 * sizeof(Key))), st));
  if (a.est_kind == 1)
    hipLaunchKernelGGL((k_mu_gather<Key, 1>), rgrid, dim3(256), 0, st, a, (const Key *)keys_sorted,
                       (const unsigned *)sc.order_sorted, total, sc.srt_prob, (double *)nullptr);
  else
    hipLaunchKernelGGL((k_mu_gather<Key, 0>), rgrid, dim3(256), 0, st, a, (const Key *)keys_sorted,
                       (const unsigned *)sc.order_sorted, total, sc.srt_prob, (double *)nullptr);
  mu_launch_apply<Key>(a, (const Key *)keys_sorted, total, st);
  return SLAMHIP_OK;
}

// The batch with the free-space fast path (map_update_kernels.h, k_mu_classify): walk, classify (the commuting
// updates are applied there), compact what is left, and sort / evaluate / apply only that.  32-bit keys.
int mu_batch_fast_tail(MuArgs &a, MuBatchScratch &sc, size_t beams, unsigned end_bit, hipStream_t st) {
  const size_t words = (((size_t)1 << (end_bit - 1)) / 32 + 4) & ~(size_t)3;  // one bit per valid key, whole uint4s
  if (words > sc.special_words) {
    SLAMHIP_CHECK(hipStreamSynchronize(st));
    SLAMHIP_CHECK(regrow(sc.special, words));
    sc.special_words = words;
  }
  if (sc.cap_beams > sc.cap_slow) {
    SLAMHIP_CHECK(hipStreamSynchronize(st));
    SLAMHIP_CHECK(regrow(sc.slow_cnt, sc.cap_beams));
    SLAMHIP_CHECK(regrow(sc.slow_off, sc.cap_beams));
    sc.cap_slow = sc.cap_beams;
  }
  hipLaunchKernelGGL(k_mu_clear_marks, dim3((unsigned)std::min<size_t>((words / 4 + 255) / 256, 4096)), dim3(256), 0, st,
                     (uint4 *)sc.special, words / 4);
  a.special = sc.special;
  a.lazy_keys = 1;
  a.walk_flag = sc.slow_cnt;  // (read by k_mu_classify before it leaves the beam's count of sorted records there)
  unsigned *keys = (unsigned *)sc.keys, *keys_c = (unsigned *)sc.keys_sorted;
  const dim3 wgrid((unsigned)((beams + 3) / 4));  // a wave per beam
  hipLaunchKernelGGL((k_mu_emit<unsigned, -1>), wgrid, dim3(256), 0, st, a, (unsigned *)nullptr);
  const dim3 cgrid((unsigned)((beams + kClassifyBeams - 1) / kClassifyBeams)), cblock(64 * kClassifyBeams);
  if (a.est_kind == 1) hipLaunchKernelGGL(k_mu_classify<1>, cgrid, cblock, 0, st, a, sc.slow_cnt);
  else hipLaunchKernelGGL(k_mu_classify<0>, cgrid, cblock, 0, st, a, sc.slow_cnt);
  {
    size_t need = 0;
    SLAMHIP_CHECK(rocprim::exclusive_scan(nullptr, need, sc.slow_cnt, sc.slow_off, 0u, beams, rocprim::plus<unsigned>(), st));
    if (need > sc.scan_temp_bytes) {
      SLAMHIP_CHECK(hipStreamSynchronize(st));
      if (sc.scan_temp) hipFree(sc.scan_temp);
      sc.scan_temp = nullptr;
      SLAMHIP_CHECK(hipMalloc(&sc.scan_temp, need));
      sc.scan_temp_bytes = need;
    }
    SLAMHIP_CHECK(rocprim::exclusive_scan(sc.scan_temp, need, sc.slow_cnt, sc.slow_off, 0u, beams, rocprim::plus<unsigned>(), st));
  }
  hipLaunchKernelGGL(k_mu_total, dim3(1), dim3(1), 0, st, (const unsigned *)sc.slow_cnt, (const unsigned *)sc.slow_off,
                     beams, sc.d_total);
  unsigned long long n_slow64 = 0;
  SLAMHIP_CHECK(hipMemcpyAsync(&n_slow64, sc.d_total, sizeof(n_slow64), hipMemcpyDeviceToHost, st));
  hipLaunchKernelGGL(k_mu_compact, wgrid, dim3(256), 0, st, a, (const unsigned *)sc.slow_cnt,
                     (const unsigned *)sc.slow_off, keys_c, sc.order_sorted);
  SLAMHIP_CHECK(hipStreamSynchronize(st));
  const unsigned n_slow = (unsigned)n_slow64;
  if (n_slow == 0) return SLAMHIP_OK;
  // the compacted records go back into the buffers the walk filled (dead by now), sorted
  size_t tb = sc.temp_bytes;
  SLAMHIP_CHECK(rocprim::radix_sort_pairs<BatchSortConfig<unsigned>>(sc.temp, tb, keys_c, keys, sc.order_sorted, sc.order,
                                                                     n_slow, 0, std::min(end_bit, 32u), st));
  a.rec_beam = sc.order;
  const dim3 sgrid((n_slow + 255) / 256);
  if (a.est_kind == 1)
    hipLaunchKernelGGL((k_mu_gather<unsigned, 1>), sgrid, dim3(256), 0, st, a, (const unsigned *)keys,
                       (const unsigned *)sc.order, n_slow, sc.srt_prob, (double *)nullptr);
  else
    hipLaunchKernelGGL((k_mu_gather<unsigned, 0>), sgrid, dim3(256), 0, st, a, (const unsigned *)keys,
                       (const unsigned *)sc.order, n_slow, sc.srt_prob, (double *)nullptr);
  mu_launch_apply<unsigned>(a, (const unsigned *)keys, n_slow, st);
  return SLAMHIP_OK;
}
}  // namespace

namespace slamhip {

// GridMapScanAdder::append_scan of ONE scan from n_jobs poses (the matched particles of a filter step),
// job k into slot slots[k] of the tile pool: one count / scan / emit / sort / gather / apply pipeline
// over all (job, beam) pairs; the sort key is (job, cell) so every cell chain of every map is
// contiguous and in beam order.  Tiles a job may write are made private first (copy-on-write).
int mu_append_batch(slamhip_ctx *ctx, TilePool *tp, const slamhip_scan_adder_cfg *cfg, int n_jobs,
                    const double *poses, const int *slots, int n, const double *range, const double *cos_a,
                    const double *sin_a, const int *is_occ, long long *n_updates_out) {
  if (!ctx || !tp || !cfg || !poses || !slots || !range || !cos_a || !sin_a) return fail("null argument");
  if (n_updates_out) *n_updates_out = 0;
  if (n_jobs <= 0 || n <= 0) return SLAMHIP_OK;
  if (cfg->rule != SLAMHIP_RULE_GMAPPING) return fail("particle maps hold GMapping cells");
  if (cfg->occupancy_estimator != 0 && cfg->occupancy_estimator != 1) return fail("unknown occupancy estimator");
  if ((long long)n_jobs * n > (1ll << 30)) return fail("batch too large");
  SLAMHIP_CHECK(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  MuBatchScratch &sc = bscratch_of(ctx);
  const double scale = tp->scale;
  const double max_range_sq = cfg->max_range * cfg->max_range;

  // job records (pose + sincos of the heading, as set_base_angle does) and the rectangle of cells each
  // job can touch, seeded with its robot cell; k_mu_count widens it by the endpoints' cells.  The record
  // count and the rectangles come back from the device in one small read (a host loop over
  // jobs x beams cost 0.5 ms per 100-particle step).
  std::vector<MuJob> jobs(n_jobs);
  std::vector<int> bbox(4 * (size_t)n_jobs);
  for (int k = 0; k < n_jobs; ++k) {
    MuJob &j = jobs[k];
    j.px = poses[3 * k];
    j.py = poses[3 * k + 1];
    ::sincos(poses[3 * k + 2], &j.sn, &j.cs);
    j.slot = slots[k];
    j.pad = 0;
    const int rcx = (int)std::floor(j.px / scale), rcy = (int)std::floor(j.py / scale);
    bbox[4 * k] = bbox[4 * k + 2] = rcx;
    bbox[4 * k + 1] = bbox[4 * k + 3] = rcy;
  }
  {  // 32-bit record offsets: a beam of range r crosses at most |dx| + |dy| + 1 <= sqrt(2) r / scale + 3 cells
    double bound = 0;
    for (int b = 0; b < n; ++b) {
      // same rule as the single-scan path: a beam that is not range-gated needs a finite end point
      if (!(std::fabs(range[b]) < 1073741824.0 * scale) && !(cfg->max_range < std::fabs(range[b])))
        return fail("a scan point that is not range-gated has a non-finite or absurdly large range");
      bound += 1.4143 * std::min(std::fabs(range[b]), cfg->max_range) / scale + 3.0;
    }
    if (bound * n_jobs >= 4.0e9) return fail("more than 2^32 cell updates in one batch: split the batch");
  }
  unsigned total = 0;

  const size_t beams = (size_t)n_jobs * n;
  if ((size_t)n > sc.cap_scan) {
    SLAMHIP_CHECK(hipStreamSynchronize(st));
    size_t cap = 2048;
    while (cap < (size_t)n) cap *= 2;
    SLAMHIP_CHECK(regrow(sc.scan, 3 * cap));
    SLAMHIP_CHECK(regrow(sc.occ, cap));
    sc.cap_scan = cap;
  }
  if (beams > sc.cap_beams) {
    SLAMHIP_CHECK(hipStreamSynchronize(st));
    size_t cap = 4096;
    while (cap < beams) cap *= 2;
    SLAMHIP_CHECK(regrow(sc.counts, cap));
    SLAMHIP_CHECK(regrow(sc.offsets, cap + 1));
    SLAMHIP_CHECK(regrow(sc.beam_end, 4 * cap));  // end point + (1/dx, 1/dy)
    SLAMHIP_CHECK(regrow(sc.beam_info, cap));
    sc.cap_beams = cap;
  }
  if ((size_t)n_jobs > sc.cap_jobs) {
    SLAMHIP_CHECK(hipStreamSynchronize(st));
    size_t cap = 64;
    while (cap < (size_t)n_jobs) cap *= 2;
    SLAMHIP_CHECK(regrow(sc.d_jobs, cap));
    SLAMHIP_CHECK(regrow(sc.d_bbox, 4 * cap));
    sc.cap_jobs = cap;
  }
  if (!sc.error_flag) SLAMHIP_CHECK(hipMalloc(&sc.error_flag, sizeof(int)));
  if (!sc.n_updates) SLAMHIP_CHECK(hipMalloc(&sc.n_updates, sizeof(unsigned long long)));
  if (!sc.h_status) SLAMHIP_CHECK(hipHostMalloc(&sc.h_status, 2 * sizeof(unsigned long long), hipHostMallocDefault));
  if (!sc.d_total) SLAMHIP_CHECK(hipMalloc(&sc.d_total, sizeof(unsigned long long)));
  auto ensure_records = [&](unsigned need_records) -> int {
    if (need_records <= sc.cap_records) return SLAMHIP_OK;
    SLAMHIP_CHECK(hipStreamSynchronize(st));
    size_t cap = 1 << 18;
    while (cap < need_records) cap *= 2;
    SLAMHIP_CHECK(regrow(sc.keys, cap));
    SLAMHIP_CHECK(regrow(sc.keys_sorted, cap));
    SLAMHIP_CHECK(regrow(sc.order, cap));
    SLAMHIP_CHECK(regrow(sc.order_sorted, cap));
    SLAMHIP_CHECK(regrow(sc.srt_prob, cap));
    if (sc.temp) hipFree(sc.temp);
    sc.temp = nullptr;
    sc.temp_bytes = 0;
    size_t tb32 = 0;  // the key buffers serve both key widths
    SLAMHIP_CHECK(rocprim::radix_sort_pairs(nullptr, sc.temp_bytes, sc.keys, sc.keys_sorted, sc.order,
                                            sc.order_sorted, cap, 0, 64, st));
    SLAMHIP_CHECK(rocprim::radix_sort_pairs<BatchSortConfig<unsigned>>(
        nullptr, tb32, (unsigned *)sc.keys, (unsigned *)sc.keys_sorted, sc.order, sc.order_sorted, cap, 0, 32, st));
    sc.temp_bytes = std::max(sc.temp_bytes, tb32);
    SLAMHIP_CHECK(hipMalloc(&sc.temp, sc.temp_bytes));
    sc.cap_records = cap;
    return SLAMHIP_OK;
  };
  const size_t cs = sc.cap_scan;
  SLAMHIP_CHECK(hipMemcpyAsync(sc.scan, range, sizeof(double) * n, hipMemcpyHostToDevice, st));
  SLAMHIP_CHECK(hipMemcpyAsync(sc.scan + cs, cos_a, sizeof(double) * n, hipMemcpyHostToDevice, st));
  SLAMHIP_CHECK(hipMemcpyAsync(sc.scan + 2 * cs, sin_a, sizeof(double) * n, hipMemcpyHostToDevice, st));
  if (is_occ) SLAMHIP_CHECK(hipMemcpyAsync(sc.occ, is_occ, sizeof(int) * n, hipMemcpyHostToDevice, st));
  SLAMHIP_CHECK(hipMemcpyAsync(sc.d_jobs, jobs.data(), sizeof(MuJob) * n_jobs, hipMemcpyHostToDevice, st));
  SLAMHIP_CHECK(hipMemcpyAsync(sc.d_bbox, bbox.data(), sizeof(int) * 4 * n_jobs, hipMemcpyHostToDevice, st));

  MuArgs a;
  std::memset(&a, 0, sizeof(a));
  a.jobs = sc.d_jobs;
  a.n_jobs = n_jobs;
  auto bind_extent = [&]() {  // (again after the extent grew)
    a.tables = tp->d_table();
    a.table_stride = tp->table_stride();
    a.tiles_x = tp->tiles_x;
    a.width = tp->width();
    a.height = tp->height();
    a.pitch = tp->width();
    a.origin_x = tp->origin_x;
    a.origin_y = tp->origin_y;
  };
  bind_extent();
  a.payload = tp->d_pool;
  a.aux = tp->d_aux;
  a.nbr_on = tp->nbr_ok ? 1 : 0;  // (in-tile masks: mu_cell_store)
  a.nbr_th = tp->nbr_th;
  a.state = tp->d_state;  // (settle states: read by k_mu_classify, kept by mu_cell_store in every batch mode)
  a.pend = tp->d_pend;    // (pending free observations: added to by k_mu_classify, folded in by mu_cell_load / _store)
  a.unknown_c0 = tp->unknown[0];
  a.fresh_ok = tp->unknown[0] < 0.0 ? 1 : 0;
  a.cell_dbl = 4;
  a.aux_stride = 2;
  a.scale = scale;
  a.range = sc.scan;
  a.cos_a = sc.scan + cs;
  a.sin_a = sc.scan + 2 * cs;
  a.is_occ = is_occ ? sc.occ : nullptr;
  a.n = n;
  a.rule = SLAMHIP_RULE_GMAPPING;
  a.est_kind = cfg->occupancy_estimator;
  a.shift_amount = cfg->area_shift_amount > 0 ? cfg->area_shift_amount : 0.01 * scale;
  a.quality = cfg->scan_quality * 1.0;
  a.base_occ_prob = cfg->base_occupied_prob;
  a.base_occ_qual = cfg->base_occupied_qual;
  a.base_empty_prob = cfg->base_empty_prob;
  a.base_empty_qual = cfg->base_empty_qual;
  a.blur = cfg->blur;
  a.max_range_sq = max_range_sq;
  a.counts = sc.counts;
  a.offsets = sc.offsets;
  a.beam_end = sc.beam_end;
  a.beam_inv = sc.beam_end + 2 * sc.cap_beams;
  a.beam_info = sc.beam_info;
  a.error_flag = sc.error_flag;
  a.n_padding = sc.n_updates;

  const dim3 bgrid((unsigned)((beams + 255) / 256));
  a.job_bbox = sc.d_bbox;
  ProfilePairGuard prof;  // slamhip_profile_read_map_update: the whole pipeline (closed on every exit)
  {
    const int prc = prof.open(ctx, st, 1);
    if (prc) return prc;
  }
  if (a.est_kind == 1) hipLaunchKernelGGL(k_mu_count<1>, bgrid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL(k_mu_count<0>, bgrid, dim3(256), 0, st, a);
  {  // offsets: device-wide exclusive scan (the one-workgroup scan of the single-scan path takes 160 us
     // for 100 x 1080 beams)
    size_t need = 0;
    SLAMHIP_CHECK(rocprim::exclusive_scan(nullptr, need, sc.counts, sc.offsets, 0u, beams, rocprim::plus<unsigned>(), st));
    if (need > sc.scan_temp_bytes) {
      SLAMHIP_CHECK(hipStreamSynchronize(st));
      if (sc.scan_temp) hipFree(sc.scan_temp);
      sc.scan_temp = nullptr;
      SLAMHIP_CHECK(hipMalloc(&sc.scan_temp, need));
      sc.scan_temp_bytes = need;
    }
    SLAMHIP_CHECK(rocprim::exclusive_scan(sc.scan_temp, need, sc.counts, sc.offsets, 0u, beams, rocprim::plus<unsigned>(), st));
  }
  hipLaunchKernelGGL(k_mu_total, dim3(1), dim3(1), 0, st, sc.counts, sc.offsets, beams, sc.d_total);
  unsigned long long total64 = 0;
  SLAMHIP_CHECK(hipMemcpyAsync(&total64, sc.d_total, sizeof(total64), hipMemcpyDeviceToHost, st));
  SLAMHIP_CHECK(hipMemcpyAsync(bbox.data(), sc.d_bbox, sizeof(int) * 4 * n_jobs, hipMemcpyDeviceToHost, st));
  SLAMHIP_CHECK(hipStreamSynchronize(st));
  if (total64 == 0) return SLAMHIP_OK;
  if (total64 >= 0xfffffff0ull) return fail("more than 2^32 cell updates in one batch: split the batch");
  total = (unsigned)total64;
  // the cells any job of the batch can touch (the walks stay inside the rectangle of their end cells); the
  // particle maps grow to hold them, like the reference's unbounded maps
  int lo_x = INT_MAX, lo_y = INT_MAX, hi_x = INT_MIN, hi_y = INT_MIN;
  for (int k = 0; k < n_jobs; ++k) {
    lo_x = std::min(lo_x, bbox[4 * k]);
    lo_y = std::min(lo_y, bbox[4 * k + 1]);
    hi_x = std::max(hi_x, bbox[4 * k + 2]);
    hi_y = std::max(hi_y, bbox[4 * k + 3]);
  }
  {
    const long long grown = tp->growths;
    const int rc_g = tile_pool_grow(tp, lo_x + tp->origin_x, lo_y + tp->origin_y, hi_x + tp->origin_x, hi_y + tp->origin_y);
    if (rc_g) return rc_g;
    if (tp->growths != grown) bind_extent();
  }
  // copy-on-write of the tiles under every job's rectangle, then room for the records
  for (int k = 0; k < n_jobs; ++k) {
    int rc = tile_pool_make_private(tp, jobs[k].slot, bbox[4 * k] + tp->origin_x, bbox[4 * k + 1] + tp->origin_y,
                                    bbox[4 * k + 2] + tp->origin_x, bbox[4 * k + 3] + tp->origin_y);
    if (rc) return rc;
  }
  int rc = tile_pool_flush(tp);
  if (rc) return rc;
  a.tables = tp->d_table();
  rc = ensure_records(total);
  if (rc) return rc;
  // key window: that rectangle with one cell of margin, in internal coordinates
  a.key_x0 = lo_x + tp->origin_x - 1;
  a.key_y0 = lo_y + tp->origin_y - 1;
  a.key_shift = 1;
  while ((1 << a.key_shift) < hi_x - lo_x + 3) ++a.key_shift;
  a.key_w = 1 << a.key_shift;  // padded to a power of two: keys decode with shift and mask
  const unsigned long long key_cells = (unsigned long long)a.key_w * (unsigned long long)(hi_y - lo_y + 3);
  unsigned cell_bits = 1, job_bits = 1;
  while ((1ull << cell_bits) < key_cells) ++cell_bits;
  while ((1u << job_bits) < (unsigned)n_jobs) ++job_bits;
  if (cell_bits + job_bits > 62) return fail("batch too large");
  a.cell_bits = (int)cell_bits;
  a.keys = sc.keys;
  a.keys_cap = (unsigned long long)sc.cap_records;
  a.rec_prob = sc.srt_prob;
  a.rec_beam = sc.order_sorted;
  // the invalid key (all ones) must still sort last: include one more bit than the valid keys use
  const unsigned end_bit = cell_bits + job_bits + 1;
  // (sorting by the cell bits alone -- one pass less, chains then ordered (cell, job) -- was measured: the
  // sort gains 100 us, k_mu_apply loses 190 us to the scattered tiles of consecutive chains)
  // free-space fast path: the const estimator's free observation must be valid and free, the bitmap of marked cells
  // at most 256 MB (end_bit <= 32)
  const bool fast = ctx->k6_batch_fast && end_bit <= 32 && cfg->base_empty_prob <= 0.5 &&
                    !std::isnan(cfg->base_empty_qual) && !ctx->k6_batch_key64;
  if (fast) {
    a.unknown_c0 = tp->unknown[0];
    a.fresh_ok = tp->unknown[0] < 0.0 ? 1 : 0;
    rc = mu_batch_fast_tail(a, sc, beams, end_bit, st);
  } else if (end_bit <= 32 && !ctx->k6_batch_key64)
    rc = mu_batch_tail<unsigned>(a, sc, total, beams, end_bit, st);
  else
    rc = mu_batch_tail<unsigned long long>(a, sc, total, beams, end_bit, st);
  if (rc) return rc;
  SLAMHIP_CHECK(hipGetLastError());
  if (prof.on()) {
    const int prc = prof.close();
    if (prc) return prc;
    ctx->prof_k6_calls += 1;
    ctx->prof_k6_records += total;
  }
  int err = 0;
  unsigned long long nu = 0;  // padding records
  rc = mu_finish(ctx, sc.error_flag, sc.n_updates, sc.h_status, &err, &nu);
  if (rc) return rc;
  nu = (unsigned long long)total - nu;
  if (n_updates_out) *n_updates_out = (long long)nu;
  if (err == 2) return fail("internal: the device counted more cell updates than the buffers hold", SLAMHIP_ERR_STATE);
  if (err)
    return fail("a beam leaves the tile extent of the particle maps: create them with a larger extent; cells "
                "inside it were updated", SLAMHIP_ERR_STATE);
  return SLAMHIP_OK;
}

// called by slamhip_ctx_destroy: the K6 scratch buffers of a context live as long as it does
void mu_release(slamhip_ctx *ctx) {
  if (ctx->mu_scratch) {
    MuScratch &s = *static_cast<MuScratch *>(ctx->mu_scratch);
    for (void *p : {(void *)s.counts, (void *)s.offsets, (void *)s.keys, (void *)s.keys_sorted, (void *)s.order,
                    (void *)s.order_sorted, (void *)s.beam_info, (void *)s.beam_end, (void *)s.scan,
                    (void *)s.srt_prob, (void *)s.srt_qual, (void *)s.occ, (void *)s.error_flag,
                    (void *)s.n_updates, s.temp, (void *)s.bins, (void *)s.offs, (void *)s.srec, s.scan_temp, (void *)s.near_bits})
      if (p) hipFree(p);
    if (s.h_status) hipHostFree(s.h_status);
    if (s.h_ring) hipHostFree(s.h_ring);
    if (s.d_ring_err) hipFree(s.d_ring_err);
    if (s.d_ring_pad) hipFree(s.d_ring_pad);
    if (s.h_offsets) hipHostFree(s.h_offsets);
    if (s.h_scan_stage) hipHostFree(s.h_scan_stage);
    if (s.h_occ_stage) hipHostFree(s.h_occ_stage);
    delete &s;
    ctx->mu_scratch = nullptr;
  }
  if (ctx->mu_bscratch) {
    MuBatchScratch &s = *static_cast<MuBatchScratch *>(ctx->mu_bscratch);
    for (void *p : {(void *)s.counts, (void *)s.offsets, (void *)s.order, (void *)s.order_sorted, (void *)s.keys,
                    (void *)s.keys_sorted, (void *)s.beam_info, (void *)s.beam_end, (void *)s.scan,
                    (void *)s.srt_prob, (void *)s.occ, (void *)s.error_flag,
                    (void *)s.d_jobs, (void *)s.d_bbox, (void *)s.n_updates, (void *)s.d_total, s.temp, s.scan_temp,
                    (void *)s.special, (void *)s.slow_cnt, (void *)s.slow_off})
      if (p) hipFree(p);
    if (s.h_status) hipHostFree(s.h_status);
    delete &s;
    ctx->mu_bscratch = nullptr;
  }
}

}  // namespace slamhip
