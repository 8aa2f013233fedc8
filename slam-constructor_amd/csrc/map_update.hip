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
// Exact scheme:
//   1. k_mu_count   one thread per beam: endpoint, range gate, upper bound |dx|+|dy|+1 of its cells
//   2. k_mu_offsets exclusive scan of the bounds (one workgroup; <= 8192 beams)
//   3. k_mu_emit    one thread per beam: the 4-connected walk with the reference's fuzzy tie rule and
//                   Bresenham fail-over; one record (cell key, observation) per cell, beam-major
//   4. rocprim::radix_sort_pairs (stable) by cell key: every cell's records end up contiguous and
//      still in beam order -- within a beam a cell is visited once, so beam order IS the
//      reference's update order for that cell
//   5. k_mu_apply   one thread per distinct cell applies its records sequentially
// HBM traffic: per (beam, cell) one 24-byte record written, sorted and read once, plus one
// read-modify-write of the cell (8-48 bytes) per distinct cell.

#include <string.h>  // rocprim's texture iterator calls ::memset without including it

#include <climits>
#include <cmath>
#include <cstring>
#include <limits>

#include <rocprim/rocprim.hpp>

#include "slamhip_internal.h"
#include "area_estimator_device.h"

namespace slamhip {

static constexpr unsigned kInvalidKey = 0xffffffffu;
static constexpr unsigned long long kInvalidKey64 = ~0ull;
static constexpr int kNuSlots = 1024;  // update counters (summed on the host)

// one scan appended from one pose into one map slot; a batch appends the SAME scan from many poses
// (the particles of the filter), each into its own copy-on-write map (tile_pool.h)
struct MuJob {
  double px, py, sn, cs;
  int slot, pad;
};

struct MuArgs {
  // batch (null: the single pose below, 32-bit keys, dense window)
  const MuJob *jobs;
  int n_jobs;
  const int *tables;  // tile tables of all slots: payload / aux are then the tile pools
  int table_stride, tiles_x, cell_bits;
  unsigned long long *keys64;
  int *job_bbox;  // per job (lo_x, lo_y, hi_x, hi_y) in external cells, reduced by k_mu_count
  // map
  double *payload;
  double *aux;
  int width, height, pitch, origin_x, origin_y, cell_dbl, aux_stride;
  double scale;
  // scan
  const double *range, *cos_a, *sin_a;
  const int *is_occ;
  int n;
  double px, py, sn, cs;  // pose, sin/cos of its heading (host sincos)
  // adder
  int rule;
  int est_kind;         // 0 ConstOccupancyEstimator, 1 AreaOccupancyEstimator
  double shift_amount;  // Q27: the estimator's function-local static (low_qual x first cell side)
  double quality, base_occ_prob, base_occ_qual, base_empty_prob, base_empty_qual, blur, max_range_sq;
  // work buffers
  unsigned *counts, *offsets;  // per beam
  unsigned *keys;
  double *rec_prob, *rec_qual;   // emit: rec_prob holds interleaved (prob, qual) pairs; apply: sorted arrays
  const double *rec_ox, *rec_oy;  // apply: the obstacle point of each sorted record
  unsigned *rec_beam;
  double *beam_end;  // 2 per beam (the obstacle point of its observations)
  int *error_flag;   // set when a touched cell lies outside the window
};

__device__ __forceinline__ bool mu_are_equal(double a, double b) {
  const double m = fmax(fabs(a), fabs(b));
  return fabs(a - b) <= 1e-7 * fmax(1.0, m);
}

// thread g of the beam kernels handles beam g % n of job g / n (a plain call is one job)
__device__ __forceinline__ MuJob mu_job(const MuArgs &a, int g) {
  if (a.jobs) return a.jobs[g / a.n];
  return MuJob{a.px, a.py, a.sn, a.cs, 0, 0};
}

__device__ __forceinline__ void mu_endpoint(const MuArgs &a, const MuJob &j, int b, double *wx, double *wy) {
  const double c = j.cs * a.cos_a[b] - j.sn * a.sin_a[b];
  const double s = j.sn * a.cos_a[b] + j.cs * a.sin_a[b];
  *wx = j.px + a.range[b] * c;
  *wy = j.py + a.range[b] * s;
}

__global__ void k_mu_count(MuArgs a) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  const bool in = g < a.n * a.n_jobs;
  unsigned cnt = 0;
  int ocx = 0, ocy = 0;
  if (in) {
    const MuJob j = mu_job(a, g);
    double wx, wy;
    mu_endpoint(a, j, g % a.n, &wx, &wy);
    a.beam_end[2 * g] = wx;
    a.beam_end[2 * g + 1] = wy;
    const double ddx = wx - j.px, ddy = wy - j.py;
    if (!(a.max_range_sq < ddx * ddx + ddy * ddy)) {
      const int rcx = (int)floor(j.px / a.scale), rcy = (int)floor(j.py / a.scale);
      ocx = (int)floor(wx / a.scale);
      ocy = (int)floor(wy / a.scale);
      cnt = (unsigned)(abs(ocx - rcx) + abs(ocy - rcy) + 1);
    }
    a.counts[g] = cnt;
  }
  if (!a.job_bbox) return;
  // cells a job can touch lie between its robot cell (host-initialised) and its endpoints: min / max of
  // the endpoint cells per job.  One atomic per wave when the wave holds a single job (a thousand
  // same-address atomics per job made this kernel 450 us), per lane otherwise.
  const int job = in ? g / a.n : -1;
  const int job0 = __shfl(job, 0, 64);
  const bool uniform = __all(job == job0 || job < 0);
  int lo_x = cnt ? ocx : INT_MAX, lo_y = cnt ? ocy : INT_MAX, hi_x = cnt ? ocx : INT_MIN, hi_y = cnt ? ocy : INT_MIN;
  if (uniform) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      lo_x = min(lo_x, __shfl_xor(lo_x, off, 64));
      lo_y = min(lo_y, __shfl_xor(lo_y, off, 64));
      hi_x = max(hi_x, __shfl_xor(hi_x, off, 64));
      hi_y = max(hi_y, __shfl_xor(hi_y, off, 64));
    }
    if ((threadIdx.x & 63) != 0) return;
  } else if (!cnt) {
    return;
  }
  if (lo_x == INT_MAX || (job0 < 0 && uniform)) return;
  int *bb = a.job_bbox + 4 * (uniform ? job0 : job);
  atomicMin(bb, lo_x);
  atomicMin(bb + 1, lo_y);
  atomicMax(bb + 2, hi_x);
  atomicMax(bb + 3, hi_y);
}

// total number of records = exclusive offset of the last beam + its count
__global__ void k_mu_total(const unsigned *counts, const unsigned *offsets, size_t beams, unsigned long long *out) {
  out[0] = (unsigned long long)offsets[beams - 1] + counts[beams - 1];
}

// exclusive scan of counts[n] by one workgroup of 1024 threads; offsets[n] = total
__global__ __launch_bounds__(1024) void k_mu_offsets(const unsigned *counts, unsigned *offsets, int n) {
  __shared__ unsigned s_part[1024];
  const int t = threadIdx.x;
  const int per = (n + 1023) / 1024;
  const int lo = t * per, hi = min(n, lo + per);
  unsigned sum = 0;
  for (int i = lo; i < hi; ++i) sum += counts[i];
  s_part[t] = sum;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const unsigned v = t >= off ? s_part[t - off] : 0;
    __syncthreads();
    s_part[t] += v;
    __syncthreads();
  }
  unsigned run = s_part[t] - sum;
  for (int i = lo; i < hi; ++i) {
    offsets[i] = run;
    run += counts[i];
  }
  if (t == 1023) offsets[n] = s_part[1023];
}

// What the walk (k_mu_emit) leaves behind per visited cell: only the sort key.  The observation itself
// (occupancy estimate, blur) is a pure function of (beam, cell) and is computed later, one thread per
// record, in k_mu_gather -- the sequential walk of a beam stays as short as its dependency chain.
__device__ __forceinline__ void mu_key(const MuArgs &a, unsigned slot, int b, int cx, int cy) {
  const int ix = cx + a.origin_x, iy = cy + a.origin_y;
  if ((unsigned)ix >= (unsigned)a.width || (unsigned)iy >= (unsigned)a.height) {
    *a.error_flag = 1;
    if (a.keys64) a.keys64[slot] = kInvalidKey64;
    else a.keys[slot] = kInvalidKey;
    return;
  }
  if (a.keys64) {  // batch: (job, cell of the virtual extent)
    const unsigned long long job = (unsigned long long)(b / a.n);
    a.keys64[slot] = (job << a.cell_bits) | ((unsigned long long)iy * (unsigned)a.width + (unsigned)ix);
  } else {
    a.keys[slot] = (unsigned)iy * (unsigned)a.pitch + (unsigned)ix;
  }
}

// per-beam quantities of WallDistanceBlurringScanAdder::handle_scan_point (grid_map_scan_adders.h:138-172)
struct MuBeam {
  double base_prob, base_qual;  // occupancy of the obstacle cell, estimated first like the reference
  double hole_dist_sq, obst_dist_sq;
  int ex, ey;  // obstacle (end) cell
};

__device__ __forceinline__ MuBeam mu_beam(const MuArgs &a, const MuJob &jb, int g) {
  MuBeam m;
  const double wx = a.beam_end[2 * g], wy = a.beam_end[2 * g + 1];
  const bool occ = a.is_occ ? a.is_occ[g % a.n] != 0 : true;
  const double scale = a.scale;
  const double d_x = wx - jb.px, d_y = wy - jb.py;
  const int bx = (int)floor(jb.px / scale), by = (int)floor(jb.py / scale);
  m.ex = (int)floor(wx / scale);
  m.ey = (int)floor(wy / scale);
  const double odx = bx - m.ex, ody = by - m.ey;
  m.obst_dist_sq = odx * odx + ody * ody;
  double blur_dist = 0;
  if (occ) {
    blur_dist = a.blur / scale;
    if (blur_dist < 0) blur_dist *= -(d_x * d_x + d_y * d_y);
  }
  m.hole_dist_sq = blur_dist * blur_dist;
  m.base_prob = occ ? a.base_occ_prob : a.base_empty_prob;
  m.base_qual = occ ? a.base_occ_qual : a.base_empty_qual;
  if (a.est_kind == 1) {
    const double base4[4] = {a.base_occ_prob, a.base_occ_qual, a.base_empty_prob, a.base_empty_qual};
    const ae::ae_rect cb{scale * m.ey, scale * (m.ey + 1), scale * m.ex, scale * (m.ex + 1)};
    const ae::ae_occ o = ae::ae_estimate(ae::ae_pt{jb.px, jb.py}, ae::ae_pt{wx, wy}, cb, occ ? 1 : 0, base4,
                                         a.shift_amount);
    m.base_prob = o.prob;
    m.base_qual = o.qual;
  }
  return m;
}

// the observation a beam makes of one of its cells: (prob, qual)
__device__ __forceinline__ double2 mu_value(const MuArgs &a, const MuJob &jb, int b, int cx, int cy, const MuBeam &bm) {
  const int ocx = bm.ex, ocy = bm.ey;
  const bool obstacle_cell = cx == ocx && cy == ocy;
  const double base_prob = bm.base_prob, base_qual = bm.base_qual;
  const double hole_dist_sq = bm.hole_dist_sq, obst_dist_sq = bm.obst_dist_sq;
  double prob, qual;
  if (obstacle_cell) {
    prob = base_prob;
    qual = base_qual;
  } else {
    prob = a.base_empty_prob;
    qual = a.base_empty_qual;
    if (a.est_kind == 1) {
      const double base4[4] = {a.base_occ_prob, a.base_occ_qual, a.base_empty_prob, a.base_empty_qual};
      const ae::ae_rect cb{a.scale * cy, a.scale * (cy + 1), a.scale * cx, a.scale * (cx + 1)};
      const ae::ae_occ o = ae::ae_estimate(ae::ae_pt{jb.px, jb.py}, ae::ae_pt{a.beam_end[2 * b], a.beam_end[2 * b + 1]},
                                           cb, 0, base4, a.shift_amount);
      prob = o.prob;
      qual = o.qual;
    }
    const double cdx = cx - ocx, cdy = cy - ocy;
    const double dist_sq = cdx * cdx + cdy * cdy;
    if (dist_sq < hole_dist_sq && hole_dist_sq < obst_dist_sq) {
      const double prob_scale = 1.0 - dist_sq / hole_dist_sq;
      prob = base_prob * prob_scale;
    }
  }
  return make_double2(prob, qual);
}

__global__ void k_mu_emit(MuArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;  // global beam index: job * n + beam
  if (b >= a.n * a.n_jobs) return;
  const unsigned cap = a.counts[b];
  if (cap == 0) return;
  const MuJob jb = mu_job(a, b);
  const unsigned base = a.offsets[b];
  const double wx = a.beam_end[2 * b], wy = a.beam_end[2 * b + 1];
  const double scale = a.scale;
  const double d_x = wx - jb.px, d_y = wy - jb.py;
  const int inc_x = 0 < d_x ? 1 : -1, inc_y = 0 < d_y ? 1 : -1;
  int px = (int)floor(jb.px / scale), py = (int)floor(jb.py / scale);
  const int bx = px, by = py;
  const int ex = (int)floor(wx / scale), ey = (int)floor(wy / scale);
  const double mid_x = (px + 0.5) * scale, mid_y = (py + 0.5) * scale;
  const double mid_cell_seg_y = d_x * jb.py + (mid_x - jb.px) * d_y;
  double e = mid_cell_seg_y - mid_y * d_x;
  const double e_x_inc = inc_x * scale * d_y;
  const double e_y_inc = -inc_y * scale * d_x;
  unsigned n = 0;
  bool failover = false;
  while (true) {
    if (n < cap) mu_key(a, base + n, b, px, py);
    ++n;
    if (px == ex && py == ey) break;
    if (cap < n) {  // fp rounding sent the walk astray: the reference restarts with Bresenham
      failover = true;
      break;
    }
    const double e_x = e + e_x_inc, e_y = e + e_y_inc;
    const double abs_err_diff = fabs(e_y) - fabs(e_x);
    if (mu_are_equal(abs_err_diff, 0)) {
      if (px == ex) py += inc_y;
      else if (py == ey) px += inc_x;
      else { px += inc_x; py += inc_y; }
      e = 0;
    } else if (0 < abs_err_diff) {
      px += inc_x;
      e = e_x;
    } else {
      py += inc_y;
      e = e_y;
    }
  }
  if (failover) {
    const int dxx = ex - bx, dyy = ey - by;
    const bool y_is_primary = abs(dxx) < abs(dyy);
    const int limit = y_is_primary ? ey : ex;
    int primary = y_is_primary ? by : bx, secondary = y_is_primary ? bx : by;
    const int d_primary = y_is_primary ? dyy : dxx, d_secondary = y_is_primary ? dxx : dyy;
    const int inc_primary = 0 < d_primary ? 1 : -1, inc_secondary = 0 < d_secondary ? 1 : -1;
    int error = 0;
    n = 0;
    while (true) {
      const int cx = y_is_primary ? secondary : primary, cy = y_is_primary ? primary : secondary;
      if (n < cap) mu_key(a, base + n, b, cx, cy);
      ++n;
      if (primary == limit) break;
      const int err_inc_primary = error + inc_primary * d_secondary;
      const int err_inc_both = err_inc_primary - inc_secondary * d_primary;
      primary += inc_primary;
      if (abs(err_inc_primary) < abs(err_inc_both)) {
        error = err_inc_primary;
      } else {
        secondary += inc_secondary;
        error = err_inc_both;
      }
    }
  }
  for (unsigned k = n; k < cap; ++k) {
    if (a.keys64) a.keys64[base + k] = kInvalidKey64;
    else a.keys[base + k] = kInvalidKey;
  }
}

__device__ __forceinline__ void mu_tbm_conj(const double *lhs, const double *rhs, double *out) {
  double tmp[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) tmp[i | j] += lhs[i] * rhs[j];
  const double tot = tmp[0] + tmp[1] + tmp[2] + tmp[3];
  if (tot == 0.0) {
    out[0] = 1.0;
    out[1] = out[2] = out[3] = 0.0;
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = tmp[i] / tot;
  }
}

// The observations in SORTED order, one thread per record: which beam made it (the walk wrote records
// beam-major, so the beam is the last b with offsets[b] <= r), which cell (from the key), then the
// occupancy estimate and blur of mu_value.  A cell's (possibly long: every beam crosses the robot's
// cell) sequential chain in k_mu_apply then streams through contiguous memory.
template <typename Key>
__global__ void k_mu_gather(MuArgs a, const Key *keys_sorted, const unsigned *order, unsigned total, int n_beams,
                            double *srt_prob, double *srt_qual, double *srt_ox, double *srt_oy) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const Key key = keys_sorted[i];
  if (key == ~Key(0)) return;  // padding of a walk that ended early: never applied
  const unsigned r = order[i];
  int lo = 0, hi = n_beams - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (a.offsets[mid] <= r) lo = mid;
    else hi = mid - 1;
  }
  const MuJob jb = mu_job(a, lo);
  int ix, iy;
  if (a.keys64) {
    const unsigned long long cellkey = (unsigned long long)key & ((1ull << a.cell_bits) - 1ull);
    ix = (int)(cellkey % (unsigned)a.width);
    iy = (int)(cellkey / (unsigned)a.width);
  } else {
    ix = (int)((unsigned)key % (unsigned)a.pitch);
    iy = (int)((unsigned)key / (unsigned)a.pitch);
  }
  const MuBeam bm = mu_beam(a, jb, lo);
  const double2 pq = mu_value(a, jb, lo, ix - a.origin_x, iy - a.origin_y, bm);
  srt_prob[i] = pq.x;
  srt_qual[i] = pq.y;
  srt_ox[i] = a.beam_end[2 * lo];
  srt_oy[i] = a.beam_end[2 * lo + 1];
}

// `a.rec_*` point at the SORTED record arrays here (k_mu_gather)
// one observation applied to one cell: the reference's `cell += aoo` for the five cell kinds
__device__ __forceinline__ void mu_apply_one(const MuArgs &a, double &c0, double &c1, double &c2, double &c3,
                                             double &x0, double &x1, double prob, double est_qual, double obx,
                                             double oby) {
  const bool invalid = isnan(prob) || isnan(est_qual);
  if (invalid && a.rule != 0) return;
  switch (a.rule) {
    case 0:  // GridCell / MockGridCell: last write wins
      c0 = prob;
      break;
    case 1:  // AffineQualityMergeCell
      c0 = (1.0 - a.quality) * c0 + a.quality * prob;
      break;
    case 2: {  // MeanProbabilityCell
      x0 += 1;
      const double that_p = 0.5 + (prob - 0.5) * a.quality;
      c0 = (c0 * (x0 - 1) + that_p) / x0;
      break;
    }
    case 3: {  // TbmBaseCell
      const double eq = est_qual * a.quality;
      const double occupied = prob * eq, empty = (1 - prob) * eq;
      const double that[4] = {1.0 - occupied - empty, empty, occupied, 0.0};
      const double cur[4] = {c0, c1, c2, c3};
      double nb[4];
      mu_tbm_conj(cur, that, nb);
      const double weight = nb[0] + nb[1] + nb[2];
      if (weight == 0.0) {
        c0 = 1.0;
        c1 = c2 = c3 = 0.0;
      } else {
        c0 = nb[0] / weight;
        c1 = nb[1] / weight;
        c2 = nb[2] / weight;
        c3 = 0.0;
      }
      break;
    }
    default: {  // GmappingBaseCell: x0 = hits, x1 = tries
      int hits = (int)x0, tries = (int)x1;
      ++tries;
      const bool is_free = prob <= 0.5;
      const double aoo_p = is_free ? 0.0 : prob;
      c0 = (c0 * (tries - 1) + aoo_p) / tries;
      if (!is_free) {
        ++hits;
        c1 = (c1 * (hits - 1) + obx) / hits;
        c2 = (c2 * (hits - 1) + oby) / hits;
      }
      x0 = hits;
      x1 = tries;
      break;
    }
  }
}

// where a sorted key's cell lives: dense window, or (job, virtual cell) -> the job's slot -> tile
template <typename Key>
__device__ __forceinline__ size_t mu_cell_index(const MuArgs &a, Key key) {
  if (!a.tables) return (size_t)key;
  const unsigned long long cellkey = (unsigned long long)key & ((1ull << a.cell_bits) - 1ull);
  const int job = (int)((unsigned long long)key >> a.cell_bits);
  const int ix = (int)(cellkey % (unsigned)a.width), iy = (int)(cellkey / (unsigned)a.width);
  const int tile = a.tables[(size_t)a.jobs[job].slot * a.table_stride + (iy >> kTileShift) * a.tiles_x + (ix >> kTileShift)];
  return ((size_t)tile << (2 * kTileShift)) + ((size_t)(iy & kTileMask) << kTileShift) + (ix & kTileMask);
}

static constexpr unsigned kLongChain = 64;  // chains at least this long go to k_mu_apply_long

__device__ __forceinline__ double mu_readlane(double v, int lane) {  // lane is wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// Long chains (the robot's own cell takes one update per beam, its neighbours hundreds): one thread
// walking such a chain pays a memory round trip per 8 records (330 us for 1080 updates).  Here the
// WAVE that holds the chain's head streams it: 64 records per coalesced load, then every lane applies
// them in order from broadcast values -- the same sequential arithmetic, executed redundantly by all
// lanes, so the result is bit-identical to the one-thread walk.
template <typename Key>
__global__ __launch_bounds__(256) void k_mu_apply_long(MuArgs a, const Key *keys, unsigned total,
                                                       unsigned long long *n_updates) {
  constexpr Key kInvalid = ~Key(0);
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  Key key = kInvalid;
  bool is_long = false;
  if (i < total) {
    key = keys[i];
    const bool head = key != kInvalid && !(i > 0 && keys[i - 1] == key);
    is_long = head && i + (kLongChain - 1) < total && keys[i + (kLongChain - 1)] == key;
  }
  unsigned long long todo = __ballot(is_long);
  while (todo) {
    const int src = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)i, src);
    const Key hkey = keys[hi];
    const size_t at = mu_cell_index<Key>(a, hkey);
    double *cell = a.payload + at * a.cell_dbl;
    double *aux = a.aux ? a.aux + at * a.aux_stride : nullptr;
    double c0 = cell[0], c1 = 0, c2 = 0, c3 = 0;
    if (a.cell_dbl == 4) {
      c1 = cell[1];
      c2 = cell[2];
      c3 = cell[3];
    }
    double x0 = aux ? aux[0] : 0, x1 = (aux && a.aux_stride > 1) ? aux[1] : 0;
    unsigned cnt = 0;
    for (unsigned j0 = hi;; j0 += 64) {
      const unsigned j = j0 + lane;
      const bool in = j < total;
      const Key k = in ? keys[j] : kInvalid;
      const double p = in ? a.rec_prob[j] : 0.0, q = in ? a.rec_qual[j] : 0.0;
      const double ox = in ? a.rec_ox[j] : 0.0, oy = in ? a.rec_oy[j] : 0.0;
      const unsigned long long m = __ballot(in && k == hkey);
      const int n_here = (m == ~0ull) ? 64 : (__ffsll((long long)~m) - 1);
      for (int t = 0; t < n_here; ++t)
        mu_apply_one(a, c0, c1, c2, c3, x0, x1, mu_readlane(p, t), mu_readlane(q, t), mu_readlane(ox, t),
                     mu_readlane(oy, t));
      cnt += (unsigned)n_here;
      if (n_here < 64) break;
    }
    if (lane == src) {
      cell[0] = c0;
      if (a.cell_dbl == 4) {
        cell[1] = c1;
        cell[2] = c2;
        cell[3] = c3;
      }
      if (aux) {
        aux[0] = x0;
        if (a.aux_stride > 1) aux[1] = x1;
      }
      atomicAdd(&n_updates[blockIdx.x & (kNuSlots - 1)], (unsigned long long)cnt);
    }
  }
}

template <typename Key>
__global__ void k_mu_apply(MuArgs a, const Key *keys, const unsigned *order, unsigned total,
                           unsigned long long *n_updates) {
  constexpr Key kInvalid = ~Key(0);
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const Key key = keys[i];
  if (key == kInvalid) return;
  if (i > 0 && keys[i - 1] == key) return;  // not the head of this cell's run
  if (i + (kLongChain - 1) < total && keys[i + (kLongChain - 1)] == key) return;  // k_mu_apply_long's
  const size_t at = mu_cell_index<Key>(a, key);
  double *cell = a.payload + at * a.cell_dbl;
  double *aux = a.aux ? a.aux + at * a.aux_stride : nullptr;
  double c0 = cell[0], c1 = 0, c2 = 0, c3 = 0;
  if (a.cell_dbl == 4) {
    c1 = cell[1];
    c2 = cell[2];
    c3 = cell[3];
  }
  double x0 = aux ? aux[0] : 0, x1 = (aux && a.aux_stride > 1) ? aux[1] : 0;
  unsigned cnt = 0;
  (void)order;
  // The chain is sequential (each update reads the previous result), but its INPUTS are not: they are
  // fetched eight records ahead so that the in-order wave pays memory latency once per chunk.
  constexpr int CH = 8;
  bool more = true;
  for (unsigned j0 = i; more && j0 < total; j0 += CH) {
    Key kk[CH];
    double pp[CH], qq[CH], oxs[CH], oys[CH];
#pragma unroll
    for (int t = 0; t < CH; ++t) {
      const unsigned j = min(j0 + t, total - 1);
      kk[t] = (j0 + t < total) ? keys[j] : kInvalid;
      pp[t] = a.rec_prob[j];
      qq[t] = a.rec_qual[j];
      oxs[t] = a.rec_ox[j];
      oys[t] = a.rec_oy[j];
    }
#pragma unroll
    for (int t = 0; t < CH; ++t) {
    if (!more) continue;
    if (kk[t] != key) {
      more = false;
      continue;
    }
    ++cnt;
    const double prob = pp[t], est_qual = qq[t], obx = oxs[t], oby = oys[t];
    mu_apply_one(a, c0, c1, c2, c3, x0, x1, prob, est_qual, obx, oby);
    }  // records of this chunk
  }    // chunks
  cell[0] = c0;
  if (a.cell_dbl == 4) {
    cell[1] = c1;
    cell[2] = c2;
    cell[3] = c3;
  }
  if (aux) {
    aux[0] = x0;
    if (a.aux_stride > 1) aux[1] = x1;
  }
  // the update count is spread over kNuSlots counters: one shared word made every wave of the grid
  // queue on the same L2 line (2.9 of 3.4 ms in a 100-particle batch, profiles/r01)
  atomicAdd(&n_updates[blockIdx.x & (kNuSlots - 1)], (unsigned long long)cnt);
}

}  // namespace slamhip

using namespace slamhip;

namespace {
struct MuScratch {
  size_t cap_records = 0, cap_beams = 0, temp_bytes = 0;
  unsigned *counts = nullptr, *offsets = nullptr, *keys = nullptr, *keys_sorted = nullptr;
  unsigned *order = nullptr, *order_sorted = nullptr, *rec_beam = nullptr;
  double *rec_prob = nullptr, *rec_qual = nullptr, *beam_end = nullptr, *scan = nullptr;
  double *srt_prob = nullptr, *srt_qual = nullptr, *srt_ox = nullptr, *srt_oy = nullptr;  // sorted records
  int *occ = nullptr, *error_flag = nullptr;
  unsigned long long *n_updates = nullptr;
  void *temp = nullptr;
  // scan re-use (see slamhip_map_append_scan)
  bool reuse_ok = false;
  const double *last_range = nullptr, *last_cos = nullptr, *last_sin = nullptr;
  const int *last_occ = nullptr;
  int last_n = -1;
};
// one scratch set per context, keyed by the context pointer (contexts are few and long-lived)
std::vector<std::pair<slamhip_ctx *, MuScratch>> g_scratch;

MuScratch &scratch_of(slamhip_ctx *ctx) {
  for (auto &p : g_scratch)
    if (p.first == ctx) return p.second;
  g_scratch.emplace_back(ctx, MuScratch{});
  return g_scratch.back().second;
}

}  // namespace

namespace slamhip {
// internal: while enabled, consecutive slamhip_map_append_scan calls that pass the very same host
// arrays upload them once (the caller guarantees their contents do not change in between)
void mu_allow_scan_reuse(slamhip_ctx *ctx, bool on) {
  MuScratch &sc = scratch_of(ctx);
  sc.reuse_ok = on;
  sc.last_n = -1;
}
}  // namespace slamhip

namespace {
__global__ void k_iota(unsigned *p, unsigned n) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = i;
}

int fail(const char *msg, int code = SLAMHIP_ERR_INVALID) {
  set_error(msg);
  return code;
}
}  // namespace

extern "C" {

int slamhip_map_append_scan(slamhip_ctx *ctx, int map_id, const slamhip_scan_adder_cfg *cfg,
                            const double pose[3], int n, const double *range, const double *cos_a,
                            const double *sin_a, const int *is_occ, long long *n_updates_out) {
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
    for (void *p : {(void *)sc.counts, (void *)sc.offsets, (void *)sc.beam_end, (void *)sc.scan, (void *)sc.occ})
      if (p) hipFree(p);
    size_t cap = 2048;
    while (cap < (size_t)n) cap *= 2;
    SLAMHIP_CHECK(hipMalloc(&sc.counts, sizeof(unsigned) * cap));
    SLAMHIP_CHECK(hipMalloc(&sc.offsets, sizeof(unsigned) * (cap + 1)));
    SLAMHIP_CHECK(hipMalloc(&sc.beam_end, sizeof(double) * 2 * cap));
    SLAMHIP_CHECK(hipMalloc(&sc.scan, sizeof(double) * 3 * cap));
    SLAMHIP_CHECK(hipMalloc(&sc.occ, sizeof(int) * cap));
    if (!sc.error_flag) SLAMHIP_CHECK(hipMalloc(&sc.error_flag, sizeof(int)));
    if (!sc.n_updates) SLAMHIP_CHECK(hipMalloc(&sc.n_updates, sizeof(unsigned long long) * kNuSlots));
    sc.cap_beams = cap;
  }
  const size_t cb = sc.cap_beams;
  // the GMapping filter appends the SAME raw scan once per particle: it brackets its loop with
  // mu_allow_scan_reuse(), and identical host arrays are then uploaded only once
  const bool reuse = sc.reuse_ok && sc.last_range == range && sc.last_cos == cos_a && sc.last_sin == sin_a &&
                     sc.last_occ == is_occ && sc.last_n == n;
  if (!reuse) {
    SLAMHIP_CHECK(hipMemcpyAsync(sc.scan, range, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
    SLAMHIP_CHECK(hipMemcpyAsync(sc.scan + cb, cos_a, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
    SLAMHIP_CHECK(hipMemcpyAsync(sc.scan + 2 * cb, sin_a, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
    if (is_occ) SLAMHIP_CHECK(hipMemcpyAsync(sc.occ, is_occ, sizeof(int) * n, hipMemcpyHostToDevice, ctx->stream));
    sc.last_range = range;
    sc.last_cos = cos_a;
    sc.last_sin = sin_a;
    sc.last_occ = is_occ;
    sc.last_n = n;
  }
  SLAMHIP_CHECK(hipMemsetAsync(sc.error_flag, 0, sizeof(int), ctx->stream));
  SLAMHIP_CHECK(hipMemsetAsync(sc.n_updates, 0, sizeof(unsigned long long) * kNuSlots, ctx->stream));

  MuArgs a;
  std::memset(&a, 0, sizeof(a));
  a.n_jobs = 1;
  a.payload = m.d_payload;
  a.aux = aux_stride ? m.d_aux : nullptr;
  a.width = m.width;
  a.height = m.height;
  a.pitch = m.pitch;
  a.origin_x = m.origin_x;
  a.origin_y = m.origin_y;
  a.cell_dbl = cell_doubles(m.cell_model);
  a.aux_stride = aux_stride;
  a.scale = m.scale;
  a.range = sc.scan;
  a.cos_a = sc.scan + cb;
  a.sin_a = sc.scan + 2 * cb;
  a.is_occ = is_occ ? sc.occ : nullptr;
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
  a.error_flag = sc.error_flag;

  const dim3 bgrid((n + 255) / 256);
  hipLaunchKernelGGL(k_mu_count, bgrid, dim3(256), 0, ctx->stream, a);
  hipLaunchKernelGGL(k_mu_offsets, dim3(1), dim3(1024), 0, ctx->stream, sc.counts, sc.offsets, n);
  // the record count is needed on the host to size the buffers: the same IEEE operations as
  // k_mu_count (no contraction on either side) give the same bounds without a device round trip
  unsigned total = 0;
  {
    const int rcx = (int)std::floor(a.px / a.scale), rcy = (int)std::floor(a.py / a.scale);
    for (int b = 0; b < n; ++b) {
      const double c = a.cs * cos_a[b] - a.sn * sin_a[b];
      const double s = a.sn * cos_a[b] + a.cs * sin_a[b];
      const double wx = a.px + range[b] * c, wy = a.py + range[b] * s;
      const double ddx = wx - a.px, ddy = wy - a.py;
      if (a.max_range_sq < ddx * ddx + ddy * ddy) continue;
      const int ocx = (int)std::floor(wx / a.scale), ocy = (int)std::floor(wy / a.scale);
      total += (unsigned)(std::abs(ocx - rcx) + std::abs(ocy - rcy) + 1);
    }
  }
  if (total == 0) {
    if (n_updates_out) *n_updates_out = 0;
    return SLAMHIP_OK;
  }
  if (total > sc.cap_records) {
    for (void *p : {(void *)sc.keys, (void *)sc.keys_sorted, (void *)sc.order, (void *)sc.order_sorted,
                    (void *)sc.rec_beam, (void *)sc.rec_prob, (void *)sc.rec_qual, sc.temp})
      if (p) hipFree(p);
    size_t cap = 1 << 16;
    while (cap < total) cap *= 2;
    SLAMHIP_CHECK(hipMalloc(&sc.keys, sizeof(unsigned) * cap));
    SLAMHIP_CHECK(hipMalloc(&sc.keys_sorted, sizeof(unsigned) * cap));
    SLAMHIP_CHECK(hipMalloc(&sc.order, sizeof(unsigned) * cap));
    SLAMHIP_CHECK(hipMalloc(&sc.order_sorted, sizeof(unsigned) * cap));
    SLAMHIP_CHECK(hipMalloc(&sc.rec_beam, sizeof(unsigned) * cap));
    SLAMHIP_CHECK(hipMalloc(&sc.rec_prob, sizeof(double) * 2 * cap));  // interleaved (prob, qual)
    SLAMHIP_CHECK(hipMalloc(&sc.rec_qual, sizeof(double) * cap));
    for (void *p : {(void *)sc.srt_prob, (void *)sc.srt_qual, (void *)sc.srt_ox, (void *)sc.srt_oy})
      if (p) hipFree(p);
    SLAMHIP_CHECK(hipMalloc(&sc.srt_prob, sizeof(double) * cap));
    SLAMHIP_CHECK(hipMalloc(&sc.srt_qual, sizeof(double) * cap));
    SLAMHIP_CHECK(hipMalloc(&sc.srt_ox, sizeof(double) * cap));
    SLAMHIP_CHECK(hipMalloc(&sc.srt_oy, sizeof(double) * cap));
    sc.temp_bytes = 0;
    SLAMHIP_CHECK(rocprim::radix_sort_pairs(nullptr, sc.temp_bytes, sc.keys, sc.keys_sorted, sc.order,
                                            sc.order_sorted, cap, 0, 32, ctx->stream));
    SLAMHIP_CHECK(hipMalloc(&sc.temp, sc.temp_bytes));
    sc.cap_records = cap;
  }
  a.keys = sc.keys;
  a.rec_prob = sc.rec_prob;
  a.rec_qual = sc.rec_qual;
  a.rec_beam = sc.rec_beam;
  hipLaunchKernelGGL(k_mu_emit, bgrid, dim3(256), 0, ctx->stream, a);
  hipLaunchKernelGGL(k_iota, dim3((total + 255) / 256), dim3(256), 0, ctx->stream, sc.order, total);
  size_t tb = sc.temp_bytes;
  // sort only the bits a cell key can occupy; the invalid key (all ones) still sorts last because
  // every valid key is < 2^nbits - 1
  unsigned nbits = 1;
  while (nbits < 32 && ((1ull << nbits) - 1) <= (unsigned long long)m.pitch * m.height) ++nbits;
  SLAMHIP_CHECK(rocprim::radix_sort_pairs(sc.temp, tb, sc.keys, sc.keys_sorted, sc.order, sc.order_sorted,
                                          total, 0, nbits, ctx->stream));
  hipLaunchKernelGGL(k_mu_gather<unsigned>, dim3((total + 255) / 256), dim3(256), 0, ctx->stream, a,
                     (const unsigned *)sc.keys_sorted, (const unsigned *)sc.order_sorted, total, n, sc.srt_prob,
                     sc.srt_qual, sc.srt_ox, sc.srt_oy);
  a.rec_prob = sc.srt_prob;
  a.rec_qual = sc.srt_qual;
  a.rec_ox = sc.srt_ox;
  a.rec_oy = sc.srt_oy;
  hipLaunchKernelGGL(k_mu_apply_long<unsigned>, dim3((total + 255) / 256), dim3(256), 0, ctx->stream, a,
                     (const unsigned *)sc.keys_sorted, total, sc.n_updates);
  hipLaunchKernelGGL(k_mu_apply<unsigned>, dim3((total + 255) / 256), dim3(256), 0, ctx->stream, a,
                     (const unsigned *)sc.keys_sorted, (const unsigned *)sc.order_sorted, total, sc.n_updates);
  SLAMHIP_CHECK(hipGetLastError());
  int err = 0;
  unsigned long long nus[kNuSlots], nu = 0;
  SLAMHIP_CHECK(hipMemcpyAsync(&err, sc.error_flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  SLAMHIP_CHECK(hipMemcpyAsync(nus, sc.n_updates, sizeof(nus), hipMemcpyDeviceToHost, ctx->stream));
  SLAMHIP_CHECK(hipStreamSynchronize(ctx->stream));
  for (int k = 0; k < kNuSlots; ++k) nu += nus[k];
  if (n_updates_out) *n_updates_out = (long long)nu;
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
  double *rec_pq = nullptr, *beam_end = nullptr, *scan = nullptr;
  double *srt_prob = nullptr, *srt_qual = nullptr, *srt_ox = nullptr, *srt_oy = nullptr;
  int *occ = nullptr, *error_flag = nullptr;
  MuJob *d_jobs = nullptr;
  int *d_bbox = nullptr;
  unsigned long long *n_updates = nullptr, *d_total = nullptr;
  void *temp = nullptr, *scan_temp = nullptr;
  size_t scan_temp_bytes = 0;
};
std::vector<std::pair<slamhip_ctx *, MuBatchScratch>> g_bscratch;

MuBatchScratch &bscratch_of(slamhip_ctx *ctx) {
  for (auto &p : g_bscratch)
    if (p.first == ctx) return p.second;
  g_bscratch.emplace_back(ctx, MuBatchScratch{});
  return g_bscratch.back().second;
}

template <typename T>
hipError_t regrow(T *&p, size_t count) {
  if (p) hipFree(p);
  p = nullptr;
  return hipMalloc(&p, sizeof(T) * count);
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
    for (int b = 0; b < n; ++b) bound += 1.4143 * std::min(std::fabs(range[b]), cfg->max_range) / scale + 3.0;
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
    SLAMHIP_CHECK(regrow(sc.beam_end, 2 * cap));
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
  if (!sc.n_updates) SLAMHIP_CHECK(hipMalloc(&sc.n_updates, sizeof(unsigned long long) * kNuSlots));
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
    SLAMHIP_CHECK(regrow(sc.rec_pq, 2 * cap));
    SLAMHIP_CHECK(regrow(sc.srt_prob, cap));
    SLAMHIP_CHECK(regrow(sc.srt_qual, cap));
    SLAMHIP_CHECK(regrow(sc.srt_ox, cap));
    SLAMHIP_CHECK(regrow(sc.srt_oy, cap));
    if (sc.temp) hipFree(sc.temp);
    sc.temp = nullptr;
    sc.temp_bytes = 0;
    SLAMHIP_CHECK(rocprim::radix_sort_pairs(nullptr, sc.temp_bytes, sc.keys, sc.keys_sorted, sc.order,
                                            sc.order_sorted, cap, 0, 64, st));
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
  SLAMHIP_CHECK(hipMemsetAsync(sc.error_flag, 0, sizeof(int), st));
  SLAMHIP_CHECK(hipMemsetAsync(sc.n_updates, 0, sizeof(unsigned long long) * kNuSlots, st));

  MuArgs a;
  std::memset(&a, 0, sizeof(a));
  a.jobs = sc.d_jobs;
  a.n_jobs = n_jobs;
  a.tables = tp->d_table();
  a.table_stride = tp->table_stride();
  a.tiles_x = tp->tiles_x;
  unsigned cell_bits = 1;
  while ((1ull << cell_bits) < (unsigned long long)tp->width() * tp->height()) ++cell_bits;
  a.cell_bits = (int)cell_bits;
  a.payload = tp->d_pool;
  a.aux = tp->d_aux;
  a.width = tp->width();
  a.height = tp->height();
  a.pitch = tp->width();
  a.origin_x = tp->origin_x;
  a.origin_y = tp->origin_y;
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
  a.error_flag = sc.error_flag;
  a.keys64 = sc.keys;
  a.rec_prob = sc.rec_pq;

  const dim3 bgrid((unsigned)((beams + 255) / 256));
  a.job_bbox = sc.d_bbox;
  hipLaunchKernelGGL(k_mu_count, bgrid, dim3(256), 0, st, a);
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
  a.keys64 = sc.keys;
  a.rec_prob = sc.rec_pq;
  hipLaunchKernelGGL(k_mu_emit, bgrid, dim3(256), 0, st, a);
  hipLaunchKernelGGL(k_iota, dim3((total + 255) / 256), dim3(256), 0, st, sc.order, total);
  unsigned job_bits = 1;
  while ((1u << job_bits) < (unsigned)n_jobs) ++job_bits;
  size_t tb = sc.temp_bytes;
  // the invalid key (all ones) must still sort last: include one more bit than the valid keys use
  const unsigned end_bit = std::min(64u, cell_bits + job_bits + 1);
  SLAMHIP_CHECK(rocprim::radix_sort_pairs(sc.temp, tb, sc.keys, sc.keys_sorted, sc.order, sc.order_sorted, total, 0,
                                          end_bit, st));
  hipLaunchKernelGGL(k_mu_gather<unsigned long long>, dim3((total + 255) / 256), dim3(256), 0, st, a,
                     (const unsigned long long *)sc.keys_sorted, (const unsigned *)sc.order_sorted, total, (int)beams,
                     sc.srt_prob, sc.srt_qual, sc.srt_ox, sc.srt_oy);
  a.rec_prob = sc.srt_prob;
  a.rec_qual = sc.srt_qual;
  a.rec_ox = sc.srt_ox;
  a.rec_oy = sc.srt_oy;
  hipLaunchKernelGGL(k_mu_apply_long<unsigned long long>, dim3((total + 255) / 256), dim3(256), 0, st, a,
                     (const unsigned long long *)sc.keys_sorted, total, sc.n_updates);
  hipLaunchKernelGGL(k_mu_apply<unsigned long long>, dim3((total + 255) / 256), dim3(256), 0, st, a,
                     (const unsigned long long *)sc.keys_sorted, (const unsigned *)sc.order_sorted, total,
                     sc.n_updates);
  SLAMHIP_CHECK(hipGetLastError());
  int err = 0;
  unsigned long long nus[kNuSlots], nu = 0;
  SLAMHIP_CHECK(hipMemcpyAsync(&err, sc.error_flag, sizeof(int), hipMemcpyDeviceToHost, st));
  SLAMHIP_CHECK(hipMemcpyAsync(nus, sc.n_updates, sizeof(nus), hipMemcpyDeviceToHost, st));
  SLAMHIP_CHECK(hipStreamSynchronize(st));
  for (int k = 0; k < kNuSlots; ++k) nu += nus[k];
  if (n_updates_out) *n_updates_out = (long long)nu;
  if (err)
    return fail("a beam leaves the tile extent of the particle maps: create them with a larger extent; cells "
                "inside it were updated", SLAMHIP_ERR_STATE);
  return SLAMHIP_OK;
}

// called by slamhip_ctx_destroy: the K6 scratch buffers of a context live as long as it does
void mu_release(slamhip_ctx *ctx) {
  for (size_t i = 0; i < g_scratch.size(); ++i) {
    if (g_scratch[i].first != ctx) continue;
    MuScratch &s = g_scratch[i].second;
    for (void *p : {(void *)s.counts, (void *)s.offsets, (void *)s.keys, (void *)s.keys_sorted, (void *)s.order,
                    (void *)s.order_sorted, (void *)s.rec_beam, (void *)s.rec_prob, (void *)s.rec_qual,
                    (void *)s.beam_end, (void *)s.scan, (void *)s.srt_prob, (void *)s.srt_qual, (void *)s.srt_ox,
                    (void *)s.srt_oy, (void *)s.occ, (void *)s.error_flag, (void *)s.n_updates, s.temp})
      if (p) hipFree(p);
    g_scratch.erase(g_scratch.begin() + i);
    break;
  }
  for (size_t i = 0; i < g_bscratch.size(); ++i) {
    if (g_bscratch[i].first != ctx) continue;
    MuBatchScratch &s = g_bscratch[i].second;
    for (void *p : {(void *)s.counts, (void *)s.offsets, (void *)s.order, (void *)s.order_sorted, (void *)s.keys,
                    (void *)s.keys_sorted, (void *)s.rec_pq, (void *)s.beam_end, (void *)s.scan, (void *)s.srt_prob,
                    (void *)s.srt_qual, (void *)s.srt_ox, (void *)s.srt_oy, (void *)s.occ, (void *)s.error_flag,
                    (void *)s.d_jobs, (void *)s.d_bbox, (void *)s.n_updates, (void *)s.d_total, s.temp, s.scan_temp})
      if (p) hipFree(p);
    g_bscratch.erase(g_bscratch.begin() + i);
    break;
  }
}

}  // namespace slamhip
