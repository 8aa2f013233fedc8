// map_update_kernels.h -- the device side of K6 (included by map_update.hip only; see its header for
// the reference lines restated and the scheme).
#pragma once

namespace slamhip {

// per-beam quantities of WallDistanceBlurringScanAdder::handle_scan_point (grid_map_scan_adders.h:138-172),
// computed once per beam by k_mu_count and read by the walk and by every record of the beam
struct MuBeam {
  int ex, ey;        // obstacle (end) cell
  double base_prob;  // occupancy of the obstacle cell, estimated first like the reference
  double base_qual;
  double hole_dist_sq, obst_dist_sq;  // (read only when the adder blurs: the first 16 bytes serve the rest)
};

struct MuArgs {
  // batch (null: the single pose below, 32-bit keys, dense window)
  const MuJob *jobs;
  int n_jobs;
  const int *tables;  // tile tables of all slots: payload / aux are then the tile pools
  int table_stride, tiles_x;
  // sort key of a record: (job << cell_bits) | cell of the key window, row-major.  The window is the
  // rectangle of internal cells (external + origin) the batch can touch -- a few metres around the
  // particles, not the map extent -- so that job and cell fit 32 bits and the radix sort runs over 4-byte
  // keys in as few passes as the window needs.  A plain call: the window is the bound map itself.
  int cell_bits, key_x0, key_y0, key_w;
  int key_shift;  // > 0: key_w = 1 << key_shift (a batch pads its window: the key decodes with shift and mask)
  void *keys;  // unsigned or unsigned long long per record (the kernels' Key parameter)
  unsigned long long keys_cap;  // records the key buffer holds: k_mu_emit refuses to write beyond it
  int *job_bbox;  // per job (lo_x, lo_y, hi_x, hi_y) in external cells, reduced by k_mu_count
  // map
  double *payload;
  double *aux;
  int width, height, pitch, origin_x, origin_y, cell_dbl, aux_stride;
  double scale;
  // scan
  const double *range, *cos_a, *sin_a;
  const int *is_occ;
  const double *beam_quality;  // per beam: the OMQE's value (plain calls only); null = 1.0 for every beam
  int n;
  double px, py, sn, cs;  // pose, sin/cos of its heading (host sincos)
  // adder
  int rule;
  int est_kind;         // 0 ConstOccupancyEstimator, 1 AreaOccupancyEstimator
  double shift_amount;  // Q27: the estimator's function-local static (low_qual x first cell side)
  double quality, base_occ_prob, base_occ_qual, base_empty_prob, base_empty_qual, blur, max_range_sq;
  // work buffers
  unsigned *counts, *offsets;  // per beam
  MuBeam *beam_info;  // per beam
  double *beam_end;   // 2 per beam (the obstacle point of its observations)
  double *beam_inv;   // 2 per beam: 1 / (end - start) per axis (area estimator's fast test, mu_free_cell_valid)
  int *error_flag;    // set when a touched cell lies outside the window (k_mu_count clears both)
  unsigned long long *n_padding;  // records that are padding or outside the map: not cell updates
  // the SORTED records (k_mu_gather -> k_mu_apply*): observation, TBM only its quality, the beam
  const double *rec_prob, *rec_qual;
  const unsigned *rec_beam;
  // plain call: the host sizes the record buffers from its own evaluation of the per-beam counts, so it also
  // knows every beam's first slot -- read from pinned memory by k_mu_emit (no scan kernel; a device-side cursor
  // bumped by 1080 waves cost 16 us of same-address atomics) and left in `offsets` for the later kernels
  const unsigned *host_offsets;
  // plain call, counting sort (see k_mu_rank): k_mu_emit counts the records of every cell of the key window
  unsigned *bins;
  unsigned n_bins;
  // the cells within near_r of the robot's cell take a record of nearly every beam: same-address atomics on their
  // bins serialize (1080 of them on the robot's own cell: 16 us), so these cells get a BITMAP of the beams that
  // visit them instead (k_mu_near_bits) -- which also is their chain in beam order
  unsigned long long *near_bits;  // [(2 near_r + 1)^2][near_words]
  int near_r, near_words, robot_ix, robot_iy;  // robot cell in internal coordinates
  // batch, free-space fast path (k_mu_classify): one bit per sort key, set by k_mu_emit for every cell a beam may
  // observe as occupied (its end cell and the cells inside the blur distance)
  unsigned *special;
  // dense GMAPPING window whose cells' pads hold neighbourhood masks for threshold nbr_th (MapView, mu_cell_store)
  int nbr_on;
  double *prob;  // TBM rule: the map's probability plane (DeviceMap::d_prob), or null -- kept by mu_cell_store
  double nbr_th;
  double unknown_c0;  // the never-observed cell's mean, if it is negative (fresh_ok)
  int fresh_ok;
  unsigned *state;  // batch: the pool's settle states (tile_pool.h), two bits per cell; null = none (dense window)
  unsigned *pend;   // batch: the pool's pending free observations per cell (tile_pool.h); null = none
  // ... with LAZY keys: a beam whose closed form holds writes no keys at all (k_mu_classify evaluates the form again,
  // step by step, instead of reading 4 bytes per record back); walk_flag[b] = 1 names the beams whose keys the
  // sequential walk wrote
  int lazy_keys;
  unsigned *walk_flag;
  // plain call, GATHER form (map_update_gather.h): one thread per cell of the key window asks the beams around its
  // direction whether their walk visits it -- no keys, no sort
  struct MuLine *lines;        // per beam: the closed form of its walk
  const unsigned *lut;         // lut[m] = beams whose pseudo-angle relative to beam 0 is below m * 4 / lut_bins
  int lut_bins, key_h;
  double rot_c, rot_s;         // cos / sin of (pose heading + angle of beam 0): world direction -> relative to beam 0
  unsigned *irr_bits;          // one word per window cell: 0, or beam + 1 of the IRREGULAR beam (sequential walk) that
                               // visits it; top bit: several do.  All zero between updates.
  unsigned char *bad;          // per beam: 1 = irregular (padded to a multiple of 8 bytes with zeros)
};

// the walk of one beam in closed form (see k_mu_emit): step k stands on the cell i_k = k - j_k steps in x and
// j_k = clamp(floor((q0 + k absA) invW), 0, k) steps in y from the robot's cell.  ok: the form was checked against
// the recurrence step by step; otherwise the beam's cells are the keys its sequential walk left in MuArgs::keys.
// The second half is what an observation of the beam needs (MuBeam without obst_dist_sq, which is the squared cell
// distance robot -> end cell and is made again from ex, ey): 64 bytes per beam, staged in LDS by k_mu_cells.
struct MuLine {
  double q0, absA, invW;
  unsigned cap;    // cells of the walk (0: the beam is range-gated)
  unsigned flags;  // 1: ok, 2: x grows, 4: y grows
  int ex, ey;
  double base_prob, base_qual, hole_dist_sq;
};
static_assert(sizeof(MuLine) == 64, "MuLine layout");

// index of a key's cell in the near grid, -1: a far cell
__device__ __forceinline__ int mu_near_index(const MuArgs &a, unsigned key) {
  const int ix = (int)(key % (unsigned)a.key_w) + a.key_x0, iy = (int)(key / (unsigned)a.key_w) + a.key_y0;
  const int nx = ix - a.robot_ix + a.near_r, ny = iy - a.robot_iy + a.near_r;
  const int side = 2 * a.near_r + 1;
  if ((unsigned)nx >= (unsigned)side || (unsigned)ny >= (unsigned)side) return -1;
  return ny * side + nx;
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

template <int EST>
__device__ __forceinline__ MuBeam mu_beam(const MuArgs &a, const MuJob &jb, int g, double wx, double wy) {
  MuBeam m;
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
  if (EST == 1) {
    const double base4[4] = {a.base_occ_prob, a.base_occ_qual, a.base_empty_prob, a.base_empty_qual};
    const ae::ae_rect cb{scale * m.ey, scale * (m.ey + 1), scale * m.ex, scale * (m.ex + 1)};
    const ae::ae_occ o = ae::ae_estimate(ae::ae_pt{jb.px, jb.py}, ae::ae_pt{wx, wy}, cb, occ ? 1 : 0, base4,
                                         a.shift_amount);
    m.base_prob = o.prob;
    m.base_qual = o.qual;
  }
  return m;
}

// (EST: occupancy estimator as a template parameter, see k_mu_gather)
template <int EST>
__global__ void k_mu_count(MuArgs a) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g == 0) {  // the status words of this update (later kernels of the stream set them)
    *a.error_flag = 0;
    *a.n_padding = 0;
  }
  if (a.near_bits) {
    const int words = (2 * a.near_r + 1) * (2 * a.near_r + 1) * a.near_words;
    for (int w = g; w < words; w += gridDim.x * blockDim.x) a.near_bits[w] = 0ull;
  }
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
    if (EST == 1) {
      a.beam_inv[2 * g] = 1.0 / ddx;
      a.beam_inv[2 * g + 1] = 1.0 / ddy;
    }
    if (!(a.max_range_sq < ddx * ddx + ddy * ddy)) {
      const int rcx = (int)floor(j.px / a.scale), rcy = (int)floor(j.py / a.scale);
      ocx = (int)floor(wx / a.scale);
      ocy = (int)floor(wy / a.scale);
      cnt = (unsigned)(abs(ocx - rcx) + abs(ocy - rcy) + 1);
      a.beam_info[g] = mu_beam<EST>(a, j, g, wx, wy);
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

// The walk of one beam as the reference runs it (RegularSquaresGrid::world_to_cells,
// regular_squares_grid.h:56-101, with the Bresenham fail-over of DiscreteSegment2D): sequential (the error
// term accumulates roundings), select-based steps, and all it leaves behind per visited cell is the sort key.
// The observation itself (occupancy estimate, blur) is a pure function of (beam, cell) and is computed later,
// one thread per record, in k_mu_gather.  Called by k_mu_emit for the beams its closed form does not settle.
template <typename KeyT>
__device__ void mu_walk_beam(const MuArgs &a, int b) {  // b = global beam index: job * n + beam
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
  const int ex = a.beam_info[b].ex, ey = a.beam_info[b].ey;
  const double mid_x = (px + 0.5) * scale, mid_y = (py + 0.5) * scale;
  const double mid_cell_seg_y = d_x * jb.py + (mid_x - jb.px) * d_y;
  double e = mid_cell_seg_y - mid_y * d_x;
  const double e_x_inc = inc_x * scale * d_y;
  const double e_y_inc = -inc_y * scale * d_x;
  const KeyT job_part = a.jobs ? (KeyT)(b / a.n) << a.cell_bits : KeyT(0);
  const unsigned row = (unsigned)a.key_w;
  const unsigned w = (unsigned)a.width, h = (unsigned)a.height;
  KeyT *out = (KeyT *)a.keys + base;
  bool bad = false;
  auto put = [&](unsigned k, int cx, int cy) {
    const unsigned ix = (unsigned)(cx + a.origin_x), iy = (unsigned)(cy + a.origin_y);
    const bool oob = ix >= w || iy >= h;
    bad |= oob;
    out[k] = oob ? ~KeyT(0) : job_part + (KeyT)(iy - (unsigned)a.key_y0) * row + (KeyT)(ix - (unsigned)a.key_x0);
  };
  // The main walk, one exit test per cell.  Position and key advance incrementally; the step is computed
  // before the exit test (and discarded with it) and built from selects, so the loop body is one basic
  // block.  A beam whose two end cells lie inside the map cannot leave it (the walk is monotone between
  // them; an astray walk is discarded), so only the others carry the per-cell bounds test.
  // The reference's loop (regular_squares_grid.h:74-98) emits at most `cap` cells, then takes one more
  // step and gives up unless that lands on the end cell: `failover` below.
  unsigned ix = (unsigned)(px + a.origin_x), iy = (unsigned)(py + a.origin_y);
  const unsigned exi = (unsigned)(ex + a.origin_x), eyi = (unsigned)(ey + a.origin_y);
  KeyT key = job_part + (KeyT)(iy - (unsigned)a.key_y0) * row + (KeyT)(ix - (unsigned)a.key_x0);
  const KeyT key_dx = (KeyT)(long long)inc_x, key_dy = (KeyT)((long long)inc_y * (long long)row);
  const unsigned uinc_x = (unsigned)inc_x, uinc_y = (unsigned)inc_y;
  unsigned n = 0;
  bool reached = false;
  auto walk = [&](auto checked) {
    constexpr bool CHECK = decltype(checked)::value;
    do {
      if (CHECK) {
        const bool oob = ix >= w || iy >= h;
        bad |= oob;
        out[n] = oob ? ~KeyT(0) : key;
      } else {
        out[n] = key;
      }
      ++n;
      const double e_x = e + e_x_inc, e_y = e + e_y_inc;
      const double abs_err_diff = fabs(e_y) - fabs(e_x);
      // are_equal(d, 0) = |d| <= 1e-7 max(1, |d|) (math_utils.h:15-25): for finite d that is |d| <= 1e-7
      const bool tie = fabs(abs_err_diff) <= 1e-7;
      const bool x_wins = 0 < abs_err_diff;
      const bool at_x = ix == exi, at_y = iy == eyi;
      reached = at_x & at_y;
      // 0 / all-ones step masks; tie: the diagonal step, degenerating to the one open axis at the end
      // row / column
      const unsigned tie_x = at_x ? 0u : ~0u, tie_y = (at_x | !at_y) ? ~0u : 0u;
      const unsigned win_x = x_wins ? ~0u : 0u;
      const unsigned mx = tie ? tie_x : win_x, my = tie ? tie_y : ~win_x;
      ix += mx & uinc_x;
      iy += my & uinc_y;
      key += ((KeyT)(long long)(int)mx & key_dx) + ((KeyT)(long long)(int)my & key_dy);
      e = tie ? 0.0 : (x_wins ? e_x : e_y);
    } while (!reached && n < cap);
  };
  if (ix < w && iy < h && exi < w && eyi < h) walk(std::false_type{});
  else walk(std::true_type{});
  // fp rounding sent the walk astray: the reference restarts with Bresenham
  const bool failover = !reached && !(ix == exi && iy == eyi);
  if (failover) {
    bad = false;  // the discarded walk touched nothing
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
      if (n < cap) put(n, cx, cy);
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
  for (unsigned k = n; k < cap; ++k) out[k] = ~KeyT(0);
  if (bad) *a.error_flag = 1;
}

template <typename Key>
__device__ __forceinline__ int mu_key_cell(const MuArgs &a, Key key, int *ix, int *iy);

// the closed form of one beam's walk (derived in k_mu_emit's comment below), shared by the kernels that evaluate it
struct MuWalkLine {
  double q0, absA, inv_W, e0, A, B;
  int bx, by, inc_x, inc_y;
};
__device__ __forceinline__ MuWalkLine mu_walk_line(const MuArgs &a, const MuJob &jb, double wx, double wy) {
  MuWalkLine L;
  const double scale = a.scale;
  const double d_x = wx - jb.px, d_y = wy - jb.py;
  L.inc_x = 0 < d_x ? 1 : -1;
  L.inc_y = 0 < d_y ? 1 : -1;
  L.bx = (int)floor(jb.px / scale);
  L.by = (int)floor(jb.py / scale);
  const double mid_x = (L.bx + 0.5) * scale, mid_y = (L.by + 0.5) * scale;
  const double mid_cell_seg_y = d_x * jb.py + (mid_x - jb.px) * d_y;
  L.e0 = mid_cell_seg_y - mid_y * d_x;
  L.A = L.inc_x * scale * d_y;
  L.B = -L.inc_y * scale * d_x;
  L.absA = fabs(L.A);
  const double absB = fabs(L.B), W = L.absA + absB;
  const double sgn = L.A < 0 ? -1.0 : 1.0;
  const double theta = (absB - L.absA) * 0.5;
  L.q0 = sgn * L.e0 - theta + absB;
  L.inv_W = 1.0 / W;
  return L;
}
// y steps among the first k steps of the walk (x steps: k - j)
// (times 1 / W rather than divided by W: k_mu_emit's check does not care how j was found, only that lane k's jn is
// lane k + 1's j -- the same expression -- and a quotient that lands on the other side of an integer sits next to a
// tie, where the walk goes to the sequential form anyway)
// (32-bit integers: a walk is far shorter than 2^31 steps, and 64-bit conversions are emulated)
__device__ __forceinline__ int mu_walk_j(double q0, double absA, double inv_W, unsigned k) {
  const double fj = floor((q0 + (double)k * absA) * inv_W);
  return (int)fmin(fmax(fj, 0.0), (double)k);
}
__device__ __forceinline__ int mu_walk_j(const MuWalkLine &L, unsigned k) { return mu_walk_j(L.q0, L.absA, L.inv_W, k); }
// a double that is the same in every lane of the wave, moved to scalar registers
__device__ __forceinline__ double mu_uniform(double v) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

// The walk of a beam the one-piece closed form does not settle, by a whole wave and exactly as mu_walk_beam would leave
// it.  What breaks the closed form is a TIE (|d| <= 1e-7: the ray passes a grid vertex): the reference then steps
// diagonally -- or along the one open axis in the end row / column -- and RESETS the error term to 0
// (regular_squares_grid.h:85-91), i.e. the rest of the walk is a new digital line from the next cell.  So the walk is
// a few closed-form pieces: the wave checks 64 steps of the current piece at a time (cell, rebuilt error term, the
// recurrence's decision against the formula's, as k_mu_emit does), emits the cells up to the first tie, starts the next
// piece behind it, and stops on the end cell; the rest of the beam's stretch is padding.  A step is classified only
// when it is clear of the tolerance by 1e-9 (the recurrence's accumulated rounding is below 1e-12); anything else --
// an unclassifiable step, a decision the formula does not reproduce, a zero-length beam, more than eight ties (a
// ray along a diagonal ties at every step), a walk that does not arrive within its cap cells (the reference's
// Bresenham fail-over) -- returns false with nothing decided, and lane 0 walks the beam step by step.
// (One scan in five has such a beam; lane 0 alone needed 25-40 us for its 600 cells.)
template <typename KeyT>
__device__ bool mu_walk_beam_wave(const MuArgs &a, int b, int lane) {
  const unsigned cap = a.counts[b];
  if (cap == 0) return true;
  const MuJob jb = mu_job(a, b);
  const MuWalkLine L = mu_walk_line(a, jb, a.beam_end[2 * b], a.beam_end[2 * b + 1]);
  const double absB = fabs(L.B);
  if (!(L.absA + absB > 0.0)) return false;  // (a beam that ends where it starts)
  const int ex = a.beam_info[b].ex, ey = a.beam_info[b].ey;
  const int steps_x = abs(ex - L.bx), steps_y = abs(ey - L.by);
  const KeyT job_part = a.jobs ? (KeyT)(b / a.n) << a.cell_bits : KeyT(0);
  const unsigned row = (unsigned)a.key_w, w = (unsigned)a.width, h = (unsigned)a.height;
  KeyT *out = (KeyT *)a.keys + a.offsets[b];
  // the current piece: first walk index, cell (in steps from the robot's) and error term there, the formula's offset
  unsigned k_base = 0u;
  int ci = 0, cj = 0;
  double e_base = L.e0, q0s = L.q0;
  bool bad = false;
  unsigned n = 0u;  // cells of the walk
  int ties = 0;
  for (unsigned k0 = 0u;;) {
    const unsigned k = k0 + (unsigned)lane, m = k - k_base;
    const int jm = mu_walk_j(q0s, L.absA, L.inv_W, m), jn = mu_walk_j(q0s, L.absA, L.inv_W, m + 1u);
    const int im = (int)m - jm;
    const int i = ci + im, j = cj + jm;
    const double e = e_base + (double)im * L.A + (double)jm * L.B;
    const double d = fabs(e + L.B) - fabs(e + L.A), ad = fabs(d);
    // 0 a plain step the formula reproduces, 1 the end cell, 2 a tie, 3 not classifiable, 4 beyond the cap
    int cls;
    if (k >= cap) cls = 4;
    else if (i == steps_x && j == steps_y) cls = 1;
    else if (i > steps_x || j > steps_y) cls = 3;
    else if (ad < 1e-7 - 1e-9) cls = 2;
    else if (ad > 1e-7 + 1e-9 && (0 < d) == (jn == jm) && jn - jm <= 1) cls = 0;
    else cls = 3;
    const unsigned long long ev = __ballot(cls != 0);
    const int first = ev ? __ffsll((long long)ev) - 1 : 64;
    const int fcls = first < 64 ? __builtin_amdgcn_readlane(cls, first) : 0;
    if (fcls == 3 || fcls == 4) return false;
    if (lane <= first) {  // (lane `first` stands on the end cell or on the cell the tie is decided from: a cell of the walk)
      const unsigned ix = (unsigned)(L.bx + L.inc_x * i + a.origin_x), iy = (unsigned)(L.by + L.inc_y * j + a.origin_y);
      const bool oob = ix >= w || iy >= h;
      bad |= oob;
      out[k] = oob ? ~KeyT(0) : job_part + (KeyT)(iy - (unsigned)a.key_y0) * row + (KeyT)(ix - (unsigned)a.key_x0);
    }
    if (first == 64) {
      k0 += 64u;
      continue;
    }
    if (fcls == 1) {
      n = k0 + (unsigned)first + 1u;
      break;
    }
    // a tie at walk index k0 + first
    if (++ties > 8) return false;
    const int ti = __builtin_amdgcn_readlane(i, first), tj = __builtin_amdgcn_readlane(j, first);
    const bool at_x = ti == steps_x, at_y = tj == steps_y;
    ci = ti + (at_x ? 0 : 1);
    cj = tj + ((at_x || !at_y) ? 1 : 0);
    k_base = k0 + (unsigned)first + 1u;
    e_base = 0.0;
    q0s = (0.0 - (absB - L.absA) * 0.5) + absB;
    k0 = k_base;
  }
  for (unsigned k = n + (unsigned)lane; k < cap; k += 64u) out[k] = ~KeyT(0);
  if (__any(bad) && lane == 0) *a.error_flag = 1;
  return true;
}

// The walk in parallel.  The reference's loop (regular_squares_grid.h:56-101) is a floating-point recurrence
// per beam -- 76 of the 170 us of a single-scan update went to 17 waves stepping 600 cells one after the other.
// But away from ties it is a plain digital line: with A = e_x_inc, B = e_y_inc (opposite signs), u = sign(A) e
// and theta = (|B| - |A|) / 2 the rule "step in x iff |e + B| > |e + A|" reads "iff u < theta", u grows by |A|
// on an x step and falls by |B| on a y step, so it stays in [theta - |B|, theta + |A|) and the number of y steps
// among the first k is
//     j_k = floor((u_0 + k |A| - theta + |B|) / (|A| + |B|)),      i_k = k - j_k .
// One wave per beam, one lane per step: lane k evaluates the cell (i_k, j_k) directly, rebuilds the error term
// e_k = e_0 + i_k A + j_k B and CHECKS the step the recurrence would take from it: the decision must be the one
// the formula takes (i_(k+1) - i_k) and must be at least 2e-7 away from the tie test (|d| <= 1e-7; the
// recurrence's accumulated rounding over a beam is below 1e-12), and the last lane must stand on the end cell.
// If every lane agrees, induction over k gives the recurrence's cell sequence; otherwise -- ties along
// diagonals, axis-parallel beams, walks that rounding sends astray -- lane 0 redoes the beam with the
// sequential walk (mu_walk_beam: tie rule, Bresenham fail-over), so the result is the same either way.
// The beam of every record (the value sorted along) is written here too (k_mu_beam_ids is gone).
// FUSE >= 0 (plain call, counting sort): k_mu_count's work for occupancy estimator FUSE is done here, by every
// lane of the beam's wave alike (one dispatch less per update; the status words are cleared by k_mu_finish of the
// update before)
template <typename KeyT, int FUSE>
__global__ __launch_bounds__(256) void k_mu_emit(MuArgs a, unsigned *beam_of) {
  if (FUSE >= 0 && a.near_bits) {
    const int words = (2 * a.near_r + 1) * (2 * a.near_r + 1) * a.near_words;
    for (int w = blockIdx.x * blockDim.x + threadIdx.x; w < words; w += gridDim.x * blockDim.x) a.near_bits[w] = 0ull;
  }
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= a.n * a.n_jobs) return;
  const int lane = threadIdx.x & 63;
  unsigned cap;
  double wx, wy;
  int ex, ey;
  if (FUSE >= 0) {
    const MuJob j = mu_job(a, b);
    mu_endpoint(a, j, b % a.n, &wx, &wy);
    const double ddx = wx - j.px, ddy = wy - j.py;
    cap = 0u;
    ex = ey = 0;
    if (!(a.max_range_sq < ddx * ddx + ddy * ddy)) {
      const int rcx = (int)floor(j.px / a.scale), rcy = (int)floor(j.py / a.scale);
      const MuBeam bm = mu_beam<(FUSE > 0 ? 1 : 0)>(a, j, b, wx, wy);
      ex = bm.ex;
      ey = bm.ey;
      cap = (unsigned)(abs(ex - rcx) + abs(ey - rcy) + 1);
      if (lane == 0) a.beam_info[b] = bm;
    }
    if (lane == 0) {
      a.beam_end[2 * b] = wx;
      a.beam_end[2 * b + 1] = wy;
      if (FUSE > 0) {
        a.beam_inv[2 * b] = 1.0 / ddx;
        a.beam_inv[2 * b + 1] = 1.0 / ddy;
      }
      a.counts[b] = cap;
    }
  } else {
    cap = a.counts[b];
  }
  if (cap == 0) return;
  const unsigned base = a.host_offsets ? a.host_offsets[b] : a.offsets[b];
  if ((unsigned long long)base + cap > a.keys_cap) {  // the host sized the buffer for another count: no write
    if (lane == 0) {
      *a.error_flag = 2;
      if (a.host_offsets) a.counts[b] = 0u;  // (k_mu_near_bits: no records of this beam)
    }
    return;
  }
  if (a.host_offsets && lane == 0) a.offsets[b] = base;
  const MuJob jb = mu_job(a, b);
  if (FUSE < 0) {
    wx = a.beam_end[2 * b];
    wy = a.beam_end[2 * b + 1];
    ex = a.beam_info[b].ex;
    ey = a.beam_info[b].ey;
  }
  const MuWalkLine L = mu_walk_line(a, jb, wx, wy);
  const double e0 = L.e0, A = L.A, B = L.B, absA = L.absA, absB = fabs(L.B);
  const int bx = L.bx, by = L.by, inc_x = L.inc_x, inc_y = L.inc_y;
  const int steps_x = abs(ex - bx), steps_y = abs(ey - by);
  const KeyT job_part = a.jobs ? (KeyT)(b / a.n) << a.cell_bits : KeyT(0);
  const unsigned row = (unsigned)a.key_w;
  const unsigned w = (unsigned)a.width, h = (unsigned)a.height;
  KeyT *out = (KeyT *)a.keys + base;
  // (an axis-parallel beam -- |A| or |B| zero -- is a digital line like any other: the per-step check below decides)
  bool ok = absA + absB > 0.0 && cap == (unsigned)(steps_x + steps_y + 1);
  bool bad = false;
  if (ok) {
    for (unsigned k0 = 0; k0 < cap; k0 += 64) {
      const unsigned k = k0 + lane;
      if (k < cap) {
        const int j = mu_walk_j(L, k), jn = mu_walk_j(L, k + 1);
        const int i = (int)k - j;
        const double e = e0 + (double)i * A + (double)j * B;
        const double d = fabs(e + B) - fabs(e + A);
        if (k + 1 < cap) {
          const bool x_formula = jn == j;  // the formula's next step is an x step
          ok = ok && fabs(d) > 2e-7 && (0 < d) == x_formula && jn - j <= 1;
        } else {
          ok = ok && i == steps_x && j == steps_y;
        }
        const unsigned ix = (unsigned)(bx + inc_x * i + a.origin_x), iy = (unsigned)(by + inc_y * j + a.origin_y);
        const bool oob = ix >= w || iy >= h;
        bad |= oob;
        if (!a.lazy_keys)
          out[k] = oob ? ~KeyT(0) : job_part + (KeyT)(iy - (unsigned)a.key_y0) * row + (KeyT)(ix - (unsigned)a.key_x0);
        if (beam_of) beam_of[base + k] = (unsigned)b;
      }
    }
  }
  ok = __all(ok);
  if (!ok) {
    // the sequential walk decides (it rewrites every key of the beam and the padding)
    if (beam_of)
      for (unsigned k = lane; k < cap; k += 64) beam_of[base + k] = (unsigned)b;
    if (FUSE >= 0) __threadfence();  // lane 0's own stores above (counts, beam_end, beam_info), read back by the walk
    if (!mu_walk_beam_wave<KeyT>(a, b, lane) && lane == 0) mu_walk_beam<KeyT>(a, b);
    __threadfence();  // the keys, read back by the whole wave below
  } else if (__any(bad) && lane == 0) {
    *a.error_flag = 1;
  }
  if (a.lazy_keys && lane == 0) a.walk_flag[b] = ok ? 0u : 1u;
  if (a.special) {
    // a SUPERSET of the cells this beam can observe with a probability above 0.5 (mu_value): its end cell and the
    // cells closer to it than the blur distance.  Along a monotone walk the L1 distance to the end cell is the
    // number of steps left, so only the last sqrt(2) * blur steps can qualify; a walk the sequential fail-over
    // rewrote is tested cell by cell.  (Lane k0 + lane reads the key it wrote itself, or evaluates the form again.)
    const double hole_sq = a.beam_info[b].hole_dist_sq;
    const unsigned m = ok ? min(cap, (unsigned)ceil(1.4143 * sqrt(hole_sq)) + 2u) : cap;
    for (unsigned k = ((cap - m) & ~63u) + lane; k < cap; k += 64) {
      if (k + m < cap) continue;
      KeyT key;
      if (a.lazy_keys && ok) {  // the closed form's cell of step k (written nowhere)
        const int j = mu_walk_j(L, k), i = (int)k - j;
        const unsigned ix = (unsigned)(bx + inc_x * i + a.origin_x), iy = (unsigned)(by + inc_y * j + a.origin_y);
        key = (ix >= w || iy >= h) ? ~KeyT(0)
                                   : job_part + (KeyT)(iy - (unsigned)a.key_y0) * row + (KeyT)(ix - (unsigned)a.key_x0);
      } else {
        key = out[k];
      }
      if (key == ~KeyT(0)) continue;
      int cix, ciy;
      mu_key_cell<KeyT>(a, key, &cix, &ciy);
      const double cdx = (cix - a.origin_x) - ex, cdy = (ciy - a.origin_y) - ey;
      const double dist_sq = cdx * cdx + cdy * cdy;
      if (dist_sq == 0.0 || dist_sq < hole_sq) atomicOr(&a.special[(size_t)(key >> 5)], 1u << (unsigned)(key & 31));
    }
  }
  if (a.bins) {
    // the beam's keys are final: count them per cell of the key window (padding and cells outside it: no bin;
    // cells near the robot: k_mu_near_bits)
    for (unsigned k = lane; k < cap; k += 64) {
      const KeyT key = out[k];
      // (a cell within near_r of the robot in both axes is at most 2 near_r steps into a walk)
      if (key < (KeyT)a.n_bins && (k > 2u * (unsigned)a.near_r || mu_near_index(a, (unsigned)key) < 0))
        atomicAdd(&a.bins[(unsigned)key], 1u);
    }
  }
}

// the internal cell (and the job) a sort key names
template <typename Key>
__device__ __forceinline__ int mu_key_cell(const MuArgs &a, Key key, int *ix, int *iy) {
  const Key cellkey = a.jobs ? key & (Key)((1ull << a.cell_bits) - 1ull) : key;
  if (a.key_shift) {
    *ix = (int)((unsigned)cellkey & ((1u << a.key_shift) - 1u)) + a.key_x0;
    *iy = (int)(cellkey >> a.key_shift) + a.key_y0;
  } else {
    *ix = (int)(cellkey % (unsigned)a.key_w) + a.key_x0;
    *iy = (int)(cellkey / (unsigned)a.key_w) + a.key_y0;
  }
  return a.jobs ? (int)(key >> a.cell_bits) : 0;
}

// Area estimator, a cell the beam passes over (is_occ = 0), cell kinds other than TBM: the estimate is always
// (base_empty.prob, some quality) unless it is INVALID -- and those cell kinds read the quality only to see
// whether it is NaN.  It is invalid in exactly one situation (area_occupancy_estimator.h:27-40,88-137): both
// ends of the beam classified outside the cell AND Rectangle::find_intersections(Segment2D) returning fewer
// than two points.  The test below proves "at least two points": the segment crosses the cell between an
// entry and an exit point that (i) lie within the segment, (ii) sit on a cell edge at least `m` away from its
// ends, so that the reference's exact interval test `e.beg <= i <= e.end` on its own (a few ulps different)
// intersection cannot fail, and (iii) are farther apart than the fuzzy point comparison can merge.  Margins are
// multiples of the tolerance of the reference's fuzzy comparisons at these coordinates (1e-7 * max(1, |x|),
// math_utils.h:15-25), which is nine orders of magnitude above the rounding of the intersection itself.
// Beams within 1000 tolerances of an axis direction, and everything the test does not prove, take the full
// estimator.  195 M records of a cfg5 step: 16.1 ms of gather with the full estimator for every record.
__device__ __forceinline__ bool mu_free_cell_valid(double x0, double y0, double x1, double y1, double inv_dx,
                                                   double inv_dy, double left, double right, double bot, double top) {
  const double dx = x1 - x0, dy = y1 - y0;
  const double big = fmax(fmax(fabs(x0), fabs(y0)), fmax(fabs(x1), fabs(y1))) + (right - left);
  const double tol = 1e-7 * fmax(1.0, big);
  if (!(fabs(dx) > 1000.0 * tol) || !(fabs(dy) > 1000.0 * tol)) return false;
  const double ta = (left - x0) * inv_dx, tb = (right - x0) * inv_dx;
  const double tc = (bot - y0) * inv_dy, td = (top - y0) * inv_dy;
  const double tx_in = fmin(ta, tb), tx_out = fmax(ta, tb), ty_in = fmin(tc, td), ty_out = fmax(tc, td);
  const double t_in = fmax(tx_in, ty_in), t_out = fmin(tx_out, ty_out);
  if (!(0.0 <= t_in && t_out <= 1.0 && t_in < t_out)) return false;
  const double m = 64.0 * tol;
  // the coordinate ALONG the edge it lies on, of the entry and of the exit point
  const bool in_through_x = tx_in > ty_in, out_through_x = tx_out < ty_out;
  const double e_along = in_through_x ? y0 + t_in * dy : x0 + t_in * dx;
  const double x_along = out_through_x ? y0 + t_out * dy : x0 + t_out * dx;
  const double e_lo = in_through_x ? bot : left, e_hi = in_through_x ? top : right;
  const double x_lo = out_through_x ? bot : left, x_hi = out_through_x ? top : right;
  if (!(e_lo + m <= e_along && e_along <= e_hi - m && x_lo + m <= x_along && x_along <= x_hi - m)) return false;
  return (t_out - t_in) * fmax(fabs(dx), fabs(dy)) > 8.0 * tol;
}

// the observation a beam makes of one of its cells: (prob, qual)
template <int EST>
__device__ __forceinline__ double2 mu_value(const MuArgs &a, int b, int cx, int cy, const MuBeam *pbm) {
  const int2 oc = *reinterpret_cast<const int2 *>(&pbm->ex);
  const int ocx = oc.x, ocy = oc.y;
  const MuBeam &bm = *pbm;
  if (cx == ocx && cy == ocy) return make_double2(bm.base_prob, bm.base_qual);
  double prob = a.base_empty_prob, qual = a.base_empty_qual;
  if (EST == 1) {
    const MuJob jb = mu_job(a, b);
    const double base4[4] = {a.base_occ_prob, a.base_occ_qual, a.base_empty_prob, a.base_empty_qual};
    const ae::ae_rect cb{a.scale * cy, a.scale * (cy + 1), a.scale * cx, a.scale * (cx + 1)};
    const double wx = a.beam_end[2 * b], wy = a.beam_end[2 * b + 1];
    if (a.rule != 3 && mu_free_cell_valid(jb.px, jb.py, wx, wy, a.beam_inv[2 * b], a.beam_inv[2 * b + 1], cb.left,
                                          cb.right, cb.bot, cb.top)) {
      // valid, and only its probability is read: (base_empty.prob, any number)
    } else {
      const ae::ae_occ o = ae::ae_estimate(ae::ae_pt{jb.px, jb.py}, ae::ae_pt{wx, wy}, cb, 0, base4, a.shift_amount);
      prob = o.prob;
      qual = o.qual;
    }
  }
  if (a.blur != 0.0) {  // (no blur: hole_dist_sq is 0 and the test below never holds)
    const double cdx = cx - ocx, cdy = cy - ocy;
    const double dist_sq = cdx * cdx + cdy * cdy;
    if (dist_sq < bm.hole_dist_sq && bm.hole_dist_sq < bm.obst_dist_sq) {
      const double prob_scale = 1.0 - dist_sq / bm.hole_dist_sq;
      prob = bm.base_prob * prob_scale;
    }
  }
  return make_double2(prob, qual);
}

// The observations in SORTED order, one thread per record: the cell comes from the key, the beam from the
// sorted value, then the occupancy estimate and blur of mu_value.  A record is 8 bytes (the observed
// probability; TBM cells also take the estimate's quality): an observation whose quality is NaN is
// dropped by every cell kind but GridCell, which never looks at it -- that is folded into a NaN
// probability here so that the chains of k_mu_apply read one double per record.
// (EST: the occupancy estimator is a template parameter -- the area estimator's rectangle clipping needs
// 208 bytes of scratch per lane, which the const estimator's instance should not carry)
template <typename Key, int EST>
__global__ void k_mu_gather(MuArgs a, const Key *keys_sorted, const unsigned *beam_sorted, unsigned total,
                            double *srt_prob, double *srt_qual) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const Key key = keys_sorted[i];
  if (key == ~Key(0)) {  // padding of a walk that ended early, or a cell outside the map: never applied
    // (counted: the number of cell updates is the number of records that are not padding -- counting
    // them in k_mu_apply instead cost one atomic per cell; padding is rare, one word takes it)
    atomicAdd(a.n_padding, 1ull);
    return;
  }
  const int b = (int)beam_sorted[i];
  int ix, iy;
  mu_key_cell<Key>(a, key, &ix, &iy);
  double2 pq = mu_value<EST>(a, b, ix - a.origin_x, iy - a.origin_y, a.beam_info + b);
  if (a.rule == 3) srt_qual[i] = pq.y;
  else if (a.rule != 0 && isnan(pq.y)) pq.x = pq.y;
  srt_prob[i] = pq.x;
}

// ---- plain call: counting sort instead of a radix sort --------------------------------------------------------
// One scan leaves ~170 k records over a key window of a few hundred thousand cells; what the update needs is every
// cell's records side by side and in beam order.  rocprim's merge sort took 8 launches / 53 us for that (half of a
// single-scan update).  Here: k_mu_emit counts the records per cell (`bins`), an exclusive scan turns the counts
// into chain starts, k_mu_scatter drops every record into its cell's chain in whatever order the atomics hand out,
// and k_mu_rank puts the chain in order -- a record's place is the number of records of its chain with a smaller
// beam (a beam visits a cell at most once, so ranks are distinct) -- and computes the observation (k_mu_gather's
// body) on the way.  Chains are a handful of records long except around the robot, whose own cell takes one of
// every beam: those cells keep a bitmap of their beams instead (MuArgs::near_bits), which gives count and rank
// without same-address atomics (first version: 16 us of serialized atomics in each of two kernels, and 57 us in
// a rank loop over 1080-record chains).  Bins are back at zero when the update is through.
// one wave per (word w of the bitmaps, step k of the walks): lane = beam 64 w + lane standing on its k-th cell; the
// lanes that stand on the same near cell are found with ballots and recorded with one atomic OR per cell (a cell is
// met at one step only, |dx| + |dy| = k, except by Bresenham fail-over walks), their number added to the cell's bin
__global__ __launch_bounds__(64) void k_mu_near_bits(MuArgs a) {
  const int w = blockIdx.x, lane = threadIdx.x;
  const unsigned k = blockIdx.y;  // 0 .. 2 near_r: a walk is monotone away from the robot's cell, later cells are far
  const int b = 64 * w + lane;
  const unsigned cap = b < a.n ? a.counts[b] : 0u;
  const unsigned key = k < cap ? ((const unsigned *)a.keys)[a.offsets[b] + k] : ~0u;
  const int idx = key < a.n_bins ? mu_near_index(a, key) : -1;
  unsigned long long todo = __ballot(idx >= 0);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const int lidx = __shfl(idx, leader, 64);
    const unsigned long long mask = __ballot(idx == lidx);
    if (lane == leader) {
      atomicOr(&a.near_bits[(size_t)lidx * a.near_words + w], mask);
      atomicAdd(&a.bins[key], (unsigned)__popcll(mask));
    }
    todo &= ~mask;
  }
}

__global__ void k_mu_scatter(MuArgs a, const unsigned *keys, const unsigned *beam_of, unsigned total, const unsigned *offs,
                             uint2 *srec) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const unsigned key = keys[i];
  if (key >= a.n_bins) return;  // padding
  const unsigned beam = beam_of[i];
  const int nidx = mu_near_index(a, key);
  unsigned pos = offs[key];
  if (nidx < 0) {
    // any free place of the chain (k_mu_rank orders it); the bin ends at 0.  Most far cells are met by one beam:
    // no atomic then
    if (offs[key + 1] - pos == 1u) a.bins[key] = 0u;
    else pos += atomicSub(&a.bins[key], 1u) - 1u;
  } else {
    // a near cell: the place IS the rank, the number of visiting beams below this one
    const unsigned long long *bw = a.near_bits + (size_t)nidx * a.near_words;
    const unsigned wq = beam >> 6;
    const unsigned long long below = (1ull << (beam & 63u)) - 1ull;
    unsigned r = 0;
#pragma unroll 4
    for (unsigned q = 0; q < (unsigned)a.near_words; ++q) {  // (uniform trip count: the loads go out together)
      const unsigned long long word = bw[q];
      r += q < wq ? (unsigned)__popcll(word) : (q == wq ? (unsigned)__popcll(word & below) : 0u);
    }
    pos += r;
  }
  srec[pos] = make_uint2(key, beam);  // one 8-byte store per record
}

template <int EST>
__global__ void k_mu_rank(MuArgs a, const uint2 *srec, const unsigned *offs, unsigned total,
                          unsigned *keys_sorted, unsigned *beam_sorted, double *srt_prob, double *srt_qual) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const unsigned valid = offs[a.n_bins];
  if (i == 0) *a.n_padding = (unsigned long long)(total - valid);
  if (i >= valid) {  // the tail the later kernels skip (k_mu_apply: a padding key heads no chain)
    keys_sorted[i] = ~0u;
    beam_sorted[i] = 0u;
    return;
  }
  const uint2 rec = srec[i];
  const unsigned key = rec.x, beam = rec.y;
  unsigned at = i;
  if (mu_near_index(a, key) < 0) {
    const unsigned start = offs[key], end = offs[key + 1];
    unsigned rank = 0;
    for (unsigned j = start; j < end; ++j) rank += srec[j].y < beam ? 1u : 0u;
    at = start + rank;
  } else {
    a.bins[key] = 0u;  // (far bins were counted down by k_mu_scatter)
  }
  keys_sorted[at] = key;
  beam_sorted[at] = beam;
  int ix, iy;
  mu_key_cell<unsigned>(a, key, &ix, &iy);
  double2 pq = mu_value<EST>(a, (int)beam, ix - a.origin_x, iy - a.origin_y, a.beam_info + beam);
  if (a.rule == 3) srt_qual[at] = pq.y;
  else if (a.rule != 0 && isnan(pq.y)) pq.x = pq.y;
  srt_prob[at] = pq.x;
}

// NO_CONFLICT (r06): the caller has established l3 = r3 = +-0 and finite, non-negative masses on both sides -- a cell's
// own conflict is +0 after every update (normalize_conflict, tbm_grid_cells.h:15-16) and an observation's always is
// (aoo2tbm, :57-66).  Seven of the conflict mass's nine products are then +-0 and adding them changes nothing (x + (+-0) =
// x for the non-negative partial sums here), so t3 = fl(fl(l1 r2) + fl(l2 r1)): the same bits, ONE dependent addition
// instead of eight -- the longest chain of the update.  (Tested per update inside the chain the shortcut LOST: 378 -> 460
// us for the robot cell's 1080 updates; established once per 64 updates by mu_wave_apply it is free.)
template <bool NO_CONFLICT = false>
__device__ __forceinline__ void mu_tbm_conj(const double *lhs, const double *rhs, double *out) {
  // tmp[i | j] += lhs[i] * rhs[j] for i, j = 0..3 in that order (transferable_belief_model.h:102-143), written out
  // per target so that nothing is indexed dynamically (the loop form left 24 bytes of scratch per lane)
  const double l0 = lhs[0], l1 = lhs[1], l2 = lhs[2], l3 = lhs[3];
  const double r0 = rhs[0], r1 = rhs[1], r2 = rhs[2], r3 = rhs[3];
  const double t0 = 0.0 + l0 * r0;
  const double t1 = ((0.0 + l0 * r1) + l1 * r0) + l1 * r1;
  const double t2 = ((0.0 + l0 * r2) + l2 * r0) + l2 * r2;
  double t3;
  if (NO_CONFLICT) t3 = l1 * r2 + l2 * r1;
  else t3 = ((((((((0.0 + l0 * r3) + l1 * r2) + l1 * r3) + l2 * r1) + l2 * r3) + l3 * r0) + l3 * r1) + l3 * r2) + l3 * r3;
  const double tot = t0 + t1 + t2 + t3;
  if (tot == 0.0) {
    out[0] = 1.0;
    out[1] = out[2] = out[3] = 0.0;
  } else {
    out[0] = t0 / tot;
    out[1] = t1 / tot;
    out[2] = t2 / tot;
    out[3] = t3 / tot;
  }
}

// the state of one cell while its chain is applied: payload c0..c3, update counters x0, x1 (kept as
// doubles: they are integers far below 2^53, so x + 1 and the products below equal the reference's
// int arithmetic converted to double)
struct MuCell {
  double c0, c1, c2, c3, x0, x1;
  unsigned pn;  // pending observations that mu_cell_load folded into x1 (a pool's cells: MuArgs::pend)
};

// TbmBaseCell::operator+= (tbm_grid_cells.h:12-19): conjunctive combination with the observation's belief, then
// normalize_conflict.  NO_CONFLICT: see mu_tbm_conj -- the caller vouches for a zero conflict mass and finite,
// non-negative masses of the cell and the observation (which the update preserves: quotients of non-negative sums).
template <bool NO_CONFLICT>
__device__ __forceinline__ void mu_step_tbm(double quality, MuCell &c, double prob, double qual) {
  if (!NO_CONFLICT && (isnan(prob) || isnan(qual))) return;
  const double eq = qual * quality;
  const double occupied = prob * eq, empty = (1 - prob) * eq;
  const double that[4] = {1.0 - occupied - empty, empty, occupied, 0.0};
  const double cur[4] = {c.c0, c.c1, c.c2, c.c3};
  double nb[4];
  mu_tbm_conj<NO_CONFLICT>(cur, that, nb);
  const double weight = nb[0] + nb[1] + nb[2];
  if (weight == 0.0) {
    c.c0 = 1.0;
    c.c1 = c.c2 = c.c3 = 0.0;
  } else {
    c.c0 = nb[0] / weight;
    c.c1 = nb[1] / weight;
    c.c2 = nb[2] / weight;
    c.c3 = 0.0;
  }
}

// one observation applied to one cell: the reference's `cell += aoo` for the five cell kinds
// (GridCell grid_cell.h:27-30, AffineQualityMergeCell / MeanProbabilityCell naive_grid_cells.h:14-20,33-40,
// TbmBaseCell tbm_grid_cells.h:57-66, GmappingBaseCell gmapping_grid_cell.h:20-33).  `qual` is read for
// TBM cells only; `obst(&x, &y)` fetches the observation's obstacle point, GMapping hits only.
// `quality` = scan quality x the observation-quality estimator's value for the record's beam (rules 1..3 read it)
template <int RULE, typename Obst>
__device__ __forceinline__ void mu_step(double quality, MuCell &c, double prob, double qual, Obst obst) {
  if (RULE == 0) {  // GridCell / MockGridCell: last write wins
    c.c0 = prob;
  } else if (RULE == 1) {  // AffineQualityMergeCell
    if (isnan(prob)) return;
    c.c0 = (1.0 - quality) * c.c0 + quality * prob;
  } else if (RULE == 2) {  // MeanProbabilityCell: x0 = _n
    if (isnan(prob)) return;
    const double n1 = c.x0 + 1;
    const double that_p = 0.5 + (prob - 0.5) * quality;
    c.c0 = (c.c0 * c.x0 + that_p) / n1;
    c.x0 = n1;
  } else if (RULE == 3) {  // TbmBaseCell
    mu_step_tbm<false>(quality, c, prob, qual);
  } else {  // GmappingBaseCell: x0 = _hits, x1 = _tries
    if (isnan(prob)) return;
    const double tries = c.x1 + 1;
    if (prob <= 0.5) {
      // a free observation of a cell whose mean is 0 leaves it at +0: (0 * k + 0) / (k + 1).  Most cells
      // of a scan are like that (free space never hit), and their chains then cost no division.
      if (c.c0 != 0.0) c.c0 = (c.c0 * c.x1 + 0.0) / tries;
      else c.c0 = 0.0;
    } else {
      c.c0 = (c.c0 * c.x1 + prob) / tries;
      double obx, oby;
      obst(&obx, &oby);
      const double hits = c.x0 + 1;
      c.c1 = (c.c1 * c.x0 + obx) / hits;
      c.c2 = (c.c2 * c.x0 + oby) / hits;
      c.x0 = hits;
    }
    c.x1 = tries;
  }
}

// where a sorted key's cell lives: dense window, or (job, virtual cell) -> the job's slot -> tile
template <typename Key>
__device__ __forceinline__ size_t mu_cell_index(const MuArgs &a, Key key) {
  if (!a.tables && !a.bins) return (size_t)key;  // the key window of a radix-sorted plain call is the bound map, pitch wide
  int ix, iy;
  if (!a.tables) {  // counting-sorted plain call: keys are cells of the window around the scan
    mu_key_cell<Key>(a, key, &ix, &iy);
    return (size_t)iy * a.pitch + ix;
  }
  const int job = mu_key_cell<Key>(a, key, &ix, &iy);
  const int tile = a.tables[(size_t)a.jobs[job].slot * a.table_stride + (iy >> kTileShift) * a.tiles_x + (ix >> kTileShift)];
  return ((size_t)tile << (2 * kTileShift)) + ((size_t)(iy & kTileMask) << kTileShift) + (ix & kTileMask);
}

// ---- batch: the free-space fast path -------------------------------------------------------------------------
// GmappingBaseCell::operator+= (gmapping_grid_cell.h:20-33) for an observation that is valid and free
// (occupancy <= 0.5) of a cell whose mean is 0:  ++_tries, mean = (0 * (_tries - 1) + 0) / _tries = +0.  No division
// result depends on the order, and _tries is an integer far below 2^53: such updates COMMUTE, a floating-point atomic
// add of 1.0 per observation gives the reference's cell bit for bit.  The same holds for a cell that was never
// observed (mean = the prototype's -1, _tries 0): its first valid free observation leaves (+0, 1).  In a scan of a
// mapped room 95+ % of the records are of that kind -- they do not need to be sorted into chains at all.
//   k_mu_emit     marks the cells a beam may observe as occupied (MuArgs::special)
//   k_mu_classify one wave per beam, over the keys of its walk: unmarked cell with mean 0 (or never observed) ->
//                 the observation's validity (area estimator) and the atomic (near the robot one per run of
//                 adjacent beams on the same cell); every other record moves to the
//                 front of the beam's own stretch of the key buffer, order kept, and is counted per beam
//   k_mu_compact  those stretches side by side (a scan of the per-beam counts gives the places), with the beam
//                 of every record, for the sort / gather / apply pipeline as before
// A cell that must go through a chain does so with ALL its records: the mark is set before, and the fast path moves
// a mean only from -1 to +0, both "fast".  The other cells may split their records between the two paths.
// (a kernel of this pipeline rather than a memset: it then shows in the pipeline's kernel trace and PMC sums)
__global__ __launch_bounds__(256) void k_mu_clear_marks(uint4 *words16, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
    words16[i] = make_uint4(0u, 0u, 0u, 0u);
}

// Sixteen adjacent beams per workgroup.  Near the robot adjacent beams cross the SAME cells -- a cell at L1 distance
// k from the robot's is met at step k of every walk that meets it, by a contiguous fan of beams (1 / (0.0058 k) of
// them at 1080 beams per turn) -- and one atomic per observation is what the kernel spends most on (they resolve in
// the fabric, 67 G/s).  For the first kNearRounds x 64 steps the sixteen waves therefore lay their settled keys side
// by side in LDS, and the first beam of every run of equal keys adds the run's length with ONE atomic.
static constexpr int kClassifyBeams = 16, kNearRounds = 3;

template <int EST>
__global__ __launch_bounds__(64 * kClassifyBeams, 8) void k_mu_classify(MuArgs a, unsigned *slow_cnt) {
  __shared__ unsigned s_key[2][kClassifyBeams][64];
  __shared__ unsigned s_max_cap;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int b = blockIdx.x * kClassifyBeams + w;
  const bool live = b < a.n * a.n_jobs;
  const unsigned cap = live ? (unsigned)__builtin_amdgcn_readfirstlane((int)a.counts[b]) : 0u;
  if (threadIdx.x == 0) s_max_cap = 0u;
  __syncthreads();
  if (lane == 0 && cap) atomicMax(&s_max_cap, cap);
  __syncthreads();
  const unsigned near_end = min((s_max_cap + 63u) & ~63u, 64u * kNearRounds);  // rounds every wave takes part in
  unsigned *out = (unsigned *)a.keys + (cap ? (unsigned)__builtin_amdgcn_readfirstlane((int)a.offsets[b]) : 0u);
  // per-beam constants of the validity proof (wave-uniform)
  MuJob jb = mu_job(a, live ? b : 0);
  jb.px = mu_uniform(jb.px);
  jb.py = mu_uniform(jb.py);
  double wx = 0, wy = 0, inv_dx = 0, inv_dy = 0;
  if (EST == 1 && cap) {
    wx = mu_uniform(a.beam_end[2 * b]);
    wy = mu_uniform(a.beam_end[2 * b + 1]);
    inv_dx = mu_uniform(a.beam_inv[2 * b]);
    inv_dy = mu_uniform(a.beam_inv[2 * b + 1]);
  }
  const long long unknown_bits = __double_as_longlong(a.unknown_c0);
  // lazy keys: the walk's cells come from the closed form k_mu_emit checked (the same expressions); only a beam the
  // sequential walk rewrote has its keys in memory
  // (what follows is the same for every lane of the beam's wave: into scalar registers, out of the way of the
  // sixty-four vector registers eight waves per SIMD leave each lane)
  const bool formula = a.lazy_keys && cap && __builtin_amdgcn_readfirstlane((int)a.walk_flag[b]) == 0;
  double q0 = 0, absA = 0, inv_W = 0;
  int bx = 0, by = 0, inc_x = 1, inc_y = 1;
  if (formula) {
    const double ewx = EST == 1 ? wx : a.beam_end[2 * b], ewy = EST == 1 ? wy : a.beam_end[2 * b + 1];
    const MuWalkLine L = mu_walk_line(a, jb, ewx, ewy);
    q0 = mu_uniform(L.q0);
    absA = mu_uniform(L.absA);
    inv_W = mu_uniform(L.inv_W);
    bx = __builtin_amdgcn_readfirstlane(L.bx);
    by = __builtin_amdgcn_readfirstlane(L.by);
    inc_x = __builtin_amdgcn_readfirstlane(L.inc_x);
    inc_y = __builtin_amdgcn_readfirstlane(L.inc_y);
  }
  const unsigned job_part = a.jobs ? (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(b / a.n) << a.cell_bits)) : 0u;
  unsigned n_slow = 0, n_pad = 0;
  for (unsigned k0 = 0; k0 < max(cap, near_end); k0 += 64) {
    const unsigned k = k0 + lane;
    unsigned key = ~0u;
    if (k < cap) {
      if (formula) {
        const int j = mu_walk_j(q0, absA, inv_W, k), i = (int)k - j;
        const unsigned ix = (unsigned)(bx + inc_x * i + a.origin_x), iy = (unsigned)(by + inc_y * j + a.origin_y);
        if (ix < (unsigned)a.width && iy < (unsigned)a.height)
          key = job_part + (iy - (unsigned)a.key_y0) * (unsigned)a.key_w + (ix - (unsigned)a.key_x0);
      } else {
        key = out[k];
      }
    }
    bool slow = false, settle = false;
    size_t at = 0;
    long long bits = 0;
    if (key == ~0u) {
      n_pad += k < cap ? 1u : 0u;
    } else if ((a.special[key >> 5] >> (key & 31u)) & 1u) {
      slow = true;
    } else {
      at = mu_cell_index<unsigned>(a, key);
      // what the cell holds: its settle state (two bits of a plane that stays in the L2) where the pool keeps one,
      // else its payload -- 32 bytes of HBM per record, a third of this kernel's traffic in cfg5
      unsigned cls;
      if (a.state) {
        cls = (a.state[at >> 4] >> (2u * (unsigned)(at & 15))) & 3u;
      } else {
        cls = mu_settle_class(a.payload[4 * at], unknown_bits, a.fresh_ok);
      }
      bits = cls == 3u ? 1ll : 0ll;  // (non-zero: the cell still holds the never-observed value)
      if (cls == 0u) {
        slow = true;
      } else {
        // const estimator: (base_empty.prob, base_empty.qual), checked by the host.  Area estimator: the record is
        // settled here only when mu_free_cell_valid PROVES the estimate valid (then it is base_empty.prob, free);
        // the others -- the robot's own cell, corner grazes -- go with the flagged ones and get the full estimator
        // in k_mu_gather: free updates commute, so a cell may take some of its records here and some there.
        if (EST == 1) {
          int ix, iy;
          mu_key_cell<unsigned>(a, key, &ix, &iy);
          const int cx = ix - a.origin_x, cy = iy - a.origin_y;
          slow = !mu_free_cell_valid(jb.px, jb.py, wx, wy, inv_dx, inv_dy, a.scale * cx, a.scale * (cx + 1),
                                     a.scale * cy, a.scale * (cy + 1));
        }
        settle = !slow;
      }
    }
    double add = 1.0;
    if (k0 < near_end) {
      // (uniform over the workgroup: every wave runs the near rounds, two buffers -> one barrier per round)
      unsigned(*sk)[64] = s_key[(k0 >> 6) & 1];
      sk[w][lane] = settle ? key : ~0u;
      __syncthreads();
      if (settle) {
        if (w > 0 && sk[w - 1][lane] == key) {
          settle = false;  // counted by the beam that heads the run
        } else {
          int len = 1;
          for (int j = w + 1; j < kClassifyBeams && sk[j][lane] == key; ++j) ++len;
          add = (double)len;
        }
      }
    }
    if (settle) {
      if (a.pend) atomicAdd(&a.pend[at], (unsigned)add);
      else unsafeAtomicAdd(&a.aux[2 * at + 1], add);
      if (bits != 0ll) {
        a.payload[4 * at] = 0.0;
        if (a.state) atomicAnd(&a.state[at >> 4], ~(2u << (2u * (unsigned)(at & 15))));  // 3 -> 1
      }
    }
    // (every key of this round was read before the first store below, and a store lands at or before its own key)
    const unsigned long long mask = __ballot(slow);
    if (slow) out[n_slow + (unsigned)__popcll(mask & ((1ull << lane) - 1ull))] = key;
    n_slow += (unsigned)__popcll(mask);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) n_pad += __shfl_xor(n_pad, off, 64);
  if (lane == 0 && live) {
    slow_cnt[b] = n_slow;
    if (n_pad) atomicAdd(a.n_padding, (unsigned long long)n_pad);
  }
}

__global__ __launch_bounds__(256) void k_mu_compact(MuArgs a, const unsigned *slow_cnt, const unsigned *slow_off,
                                                    unsigned *keys_out, unsigned *beam_out) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= a.n * a.n_jobs) return;
  const unsigned n = slow_cnt[b];
  if (n == 0) return;
  const unsigned *in = (const unsigned *)a.keys + a.offsets[b];
  const unsigned to = slow_off[b];
  for (unsigned j = threadIdx.x & 63; j < n; j += 64) {
    keys_out[to + j] = in[j];
    beam_out[to + j] = (unsigned)b;
  }
}

template <int RULE>
__device__ __forceinline__ MuCell mu_cell_load(const MuArgs &a, size_t at) {
  MuCell c{0, 0, 0, 0, 0, 0};
  if (RULE >= 3) {  // 4 doubles per cell: one 32-byte load
    const double4 v = reinterpret_cast<const double4 *>(a.payload)[at];
    c.c0 = v.x;
    c.c1 = v.y;
    c.c2 = v.z;
    c.c3 = v.w;
  } else {
    c.c0 = a.payload[at];
  }
  if (RULE == 2) c.x0 = a.aux[at];
  if (RULE == 4) {
    const double2 x = reinterpret_cast<const double2 *>(a.aux)[at];
    c.x0 = x.x;
    c.x1 = x.y;
    if (a.pend) {  // tries = the counter + what the fast path has settled since it was last written
      c.pn = a.pend[at];
      c.x1 = c.x1 + (double)c.pn;
    }
  }
  return c;
}

// `was` is what mu_cell_load returned.  A GMapping cell in free space (mean 0) only counts tries: its 32-byte
// payload keeps its bits and is not written back -- most cells of a scan are like that, and with one cell per
// 32-byte sector that write is a third of the kernel's HBM bytes.
template <int RULE>
__device__ __forceinline__ void mu_cell_store(const MuArgs &a, size_t at, const MuCell &c, const MuCell &was) {
  if (RULE == 4 && a.state) {
    // the pool's settle states follow the mean (bits are only ever cleared: tile_pool.h)
    const long long ub = __double_as_longlong(a.unknown_c0);
    const unsigned now = mu_settle_class(c.c0, ub, a.fresh_ok), before = mu_settle_class(was.c0, ub, a.fresh_ok);
    const unsigned gone = before & ~now;
    if (gone) atomicAnd(&a.state[at >> 4], ~(gone << (2u * (unsigned)(at & 15))));
  }
  if (RULE == 4) {
    const bool same = __double_as_longlong(c.c0) == __double_as_longlong(was.c0) &&
                      __double_as_longlong(c.c1) == __double_as_longlong(was.c1) &&
                      __double_as_longlong(c.c2) == __double_as_longlong(was.c2) &&
                      __double_as_longlong(c.c3) == __double_as_longlong(was.c3);
    if (!same) {
      if (a.nbr_on) {
        // the pad holds this cell's neighbourhood mask, which the neighbours' threads change with atomics while this
        // one stores: the pad is not written here.  A cell that changes sides flips its bit in the nine masks it is in
        // (the cell at (+dx, +dy) sees this one at (-dx, -dy): bit 8 - i).
        reinterpret_cast<double2 *>(a.payload)[2 * at] = make_double2(c.c0, c.c1);
        a.payload[4 * at + 2] = c.c2;
        if ((c.c0 < a.nbr_th) != (was.c0 < a.nbr_th)) {
          if (a.tables) {  // a tile of a pool: the masks know the cells of their own tile only (tile_pool.h)
            const int lx = (int)(at & kTileMask), ly = (int)((at >> kTileShift) & kTileMask);
            const size_t base = at & ~(size_t)(kTileCells - 1);
#pragma unroll
            for (int i = 0; i < 9; ++i) {
              const int x = lx + i / 3 - 1, y = ly + i % 3 - 1;
              if ((unsigned)x < (unsigned)kTileSide && (unsigned)y < (unsigned)kTileSide)
                atomicXor(reinterpret_cast<unsigned *>(a.payload + 4 * (base + ((size_t)y << kTileShift) + x) + 3), 1u << (8 - i));
            }
          } else {
            const int iy = (int)(at / (size_t)a.pitch), ix = (int)(at - (size_t)iy * a.pitch);
#pragma unroll
            for (int i = 0; i < 9; ++i) {
              const int x = ix + i / 3 - 1, y = iy + i % 3 - 1;
              if ((unsigned)x < (unsigned)a.width && (unsigned)y < (unsigned)a.height)
                atomicXor(reinterpret_cast<unsigned *>(a.payload + 4 * ((size_t)y * a.pitch + x) + 3), 1u << (8 - i));
            }
          }
        }
      } else {
        reinterpret_cast<double4 *>(a.payload)[at] = make_double4(c.c0, c.c1, c.c2, c.c3);
      }
    }
  } else if (RULE == 3) {
    reinterpret_cast<double4 *>(a.payload)[at] = make_double4(c.c0, c.c1, c.c2, c.c3);
    // the scorers' per-beam probability of this cell, where the map keeps a plane of them (the same function of the
    // same four doubles the scorers would evaluate per (pose, beam): score_device.h cell_probability<TBM>)
    if (a.prob) a.prob[at] = tbm_discrepancy_probability(c.c0, c.c1, c.c2, c.c3);
  } else {
    a.payload[at] = c.c0;
  }
  if (RULE == 2) a.aux[at] = c.x0;
  if (RULE == 4) {
    reinterpret_cast<double2 *>(a.aux)[at] = make_double2(c.x0, c.x1);
    if (was.pn) a.pend[at] = 0u;  // (folded into x1 by the load; no fast path runs beside a cell store)
  }
}

__device__ __forceinline__ double mu_readlane(double v, int lane) {  // lane is wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// x / y, correctly rounded, from a reciprocal of y refined AHEAD of x: the IEEE division sequence the compiler emits
// for a / b (v_div_scale, v_rcp, two Newton steps, quotient, residual, v_div_fmas, v_div_fixup) splits into a part that
// needs the denominator alone and a tail of three dependent operations on the numerator.  Inside 2^-500 .. 2^500 the
// scaling steps leave both operands alone, so the refined reciprocal can be made a step early; outside (and for zeros,
// infinities, NaNs) the plain division runs.  tools/probes/div_probe.hip: 16.7 M divisions, all bit-equal.
__device__ __forceinline__ double mu_refined_rcp(double y) {
  const double r0 = __builtin_amdgcn_rcp(y);
  const double f0 = __builtin_fma(-y, r0, 1.0);
  const double r1 = __builtin_fma(r0, f0, r0);
  const double f2 = __builtin_fma(-y, r1, 1.0);
  return __builtin_fma(r1, f2, r1);
}
// x / y from y's refined reciprocal with NO scaling or fix-up step: exact where those steps do nothing (the callers
// establish the operand ranges; tools/probes/div_probe.hip, tbm_div_probe.hip)
__device__ __forceinline__ double mu_div_nofix(double x, double y, double r) {
  const double q = x * r;
  const double e = __builtin_fma(-y, q, x);
  return __builtin_fma(e, r, q);
}
__device__ __forceinline__ bool mu_div_safe(double x) {
  const double ax = fabs(x);
  return ax > 0x1p-500 && ax < 0x1p500;
}
__device__ __forceinline__ double mu_div_with(double x, double y, double r) {
  if (!(mu_div_safe(x) && mu_div_safe(y))) return x / y;
  const double q = x * r;
  const double e = __builtin_fma(-y, q, x);
  const double q2 = __builtin_fma(e, r, q);
  return __builtin_amdgcn_div_fixup(q2, y, x);
}

// Up to 64 observations of ONE cell applied in order by a whole wave: lane t < n_here holds record t (probability p,
// TBM quality q, update quality ql, obstacle point ox / oy for GMapping hits); every lane ends with the same cell
// state -- the sequential arithmetic executed redundantly from lane broadcasts, bit-identical to one thread walking
// the chain.  GMapping cells: a run of free observations of a cell whose mean is 0 only counts tries (see mu_step)
// and is skipped in one step from the ballot of the hits.  MeanProbabilityCell: the division's reciprocal is made a
// step ahead (see below).
template <int RULE>
__device__ __forceinline__ void mu_wave_apply(const MuArgs &a, MuCell &c, int lane, int n_here, bool in, double p,
                                              double q, double ql, double ox, double oy, double *s_wave = nullptr) {
  constexpr bool kReadsQuality = RULE >= 1 && RULE <= 3;
  unsigned long long busy = ~0ull;  // records that need arithmetic
  if (RULE == 4) busy = __ballot(in && !(p <= 0.5));  // hits and NaNs
  int t = 0;
  if (RULE == 2) {
    // MeanProbabilityCell: c = (c n + p') / (n + 1), a division per observation that waits for the one before --
    // 1080 of them on the robot's own cell, 100 ns each, were the whole kernel.  The denominator is known a
    // step early: its refined reciprocal is made next to the step before (two independent chains in one
    // loop body), which leaves three dependent operations of the division behind the numerator.
    // The round takes this form only when every operand stays inside the range where the division's scaling
    // steps do nothing (all p' in 2^-400 .. 2^400, none NaN, the mean non-negative and below 2^400: the mean is
    // then a convex combination of such values all along); otherwise mu_step's plain divisions below.
    const double tp = 0.5 + (p - 0.5) * ql;  // mu_step's that_p, per lane
    const bool mine = lane < n_here;
    const bool fine = !mine || (tp > 0x1p-400 && tp < 0x1p400);
    if (__all(fine) && c.c0 >= 0.0 && c.c0 < 0x1p400 && c.x0 >= 0.0 && c.x0 < 0x1p52 && s_wave) {
      // With 192 doubles of LDS of the wave's own (k_mu_cells): everything a step needs besides the running mean --
      // its denominator n + 1 + t (exact: integers below 2^53), that denominator's refined reciprocal, the
      // observation -- is made by lane t for step t, all 64 at once, and the loop reads it back with uniform LDS
      // loads that depend on nothing in the chain.  What is left per observation is the chain itself: multiply, add,
      // multiply, two fused multiply-adds.  (From lane broadcasts and with the reciprocal made inside the loop a
      // step took 65 ns: fourteen vector instructions, a quarter-rate v_rcp among them, for five that depend.)
      const double nj = c.x0 + 1.0 + (double)lane;
      s_wave[lane] = mu_refined_rcp(nj);
      s_wave[64 + lane] = nj;
      s_wave[128 + lane] = tp;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      double c0 = c.c0, x0 = c.x0;
#pragma unroll 4
      for (; t < n_here; ++t) {
        const double r = s_wave[t], n1 = s_wave[64 + t], tpt = s_wave[128 + t];
        const double x = c0 * x0 + tpt;
        const double qq = x * r;
        const double e = __builtin_fma(-n1, qq, x);
        c0 = __builtin_fma(e, r, qq);
        x0 = n1;
      }
      c.c0 = c0;
      c.x0 = x0;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // these loads before the caller's next stores
      __builtin_amdgcn_wave_barrier();
    } else if (__all(fine) && c.c0 >= 0.0 && c.c0 < 0x1p400 && c.x0 >= 0.0 && c.x0 < 0x1p52) {
      double r = mu_refined_rcp(c.x0 + 1.0);
      for (; t < n_here; ++t) {
        const double n1 = c.x0 + 1;
        const double x = c.c0 * c.x0 + mu_readlane(tp, t);
        const double r_next = mu_refined_rcp(n1 + 1.0);  // (independent of x: scheduled beside the tail below)
        const double qq = x * r;
        const double e = __builtin_fma(-n1, qq, x);
        // (no v_div_fixup: with both operands inside 2^-500 .. 2^500 -- the test above -- it has nothing to fix, and
        // it was one of six dependent operations per observation; tools/probes/div_probe.hip: 0 of 13.6 M different)
        c.c0 = __builtin_fma(e, r, qq);
        c.x0 = n1;
        r = r_next;
      }
    }
  }
  if (RULE == 3) {
    // TbmBaseCell chains (r06).  The robot's own cell takes one update per beam -- 1080 of them, one after the other,
    // 350 ns each: conjunctive combination (the conflict mass alone eight dependent additions), normalize (four
    // divisions by the total), normalize_conflict (three by the weight).  A round of up to 64 observations first
    // establishes, each lane looking at its own observation and all at the cell,
    //   * observation: unknown mass >= 0.5, empty and occupied mass +0 or in [2^-40, 1] (aoo2tbm of any probability and
    //     quality in [0, 1] with quality <= 0.5), conflict +0 by construction;
    //   * cell: conflict +-0 ... +0, the other masses +0 or in [2^-300, 2], their sum in [0.5, 2];
    // which every update of the round preserves well enough (a mass shrinks by at most ~2^-2.2 per update: 2^-141 over a
    // round; sums stay within a few ulps of 1 after the first) for the round to run WITHOUT per-update tests:
    //   * conflict mass t3 = fl(fl(l1 r2) + fl(l2 r1)): seven of its nine products are +0 and adding +0 to a
    //     non-negative sum changes nothing -- the same bits, one dependent addition (the leading `0.0 +` of every mass
    //     likewise);
    //   * the quotients from two reciprocals refined ahead, q = x r, e = fma(-y, q, x), q' = fma(e, r, q) -- the IEEE
    //     division sequence without its scaling and fix-up steps, which do nothing for numerators 0 or >= 2^-600 over
    //     denominators in [0.2, 2.5] (tools/probes/tbm_div_probe.hip: 33.5 M divisions, all bit-equal); total and weight
    //     are >= 0.45 here, so neither the `== 0` branches nor the unused fourth quotient (the conflict share) exist.
    // Anything else takes mu_step's plain form below.  (The same tests made PER UPDATE lost: 378 -> 460 us for the robot's
    // cell; per round they are free: 378 -> 329 with the conflict shortcut alone, -> LOG r06 with the quotients.)
    const double qmine = (kReadsQuality && a.beam_quality) ? ql : a.quality;
    const double eq = q * qmine;
    const double occupied = p * eq, empty = (1 - p) * eq, unknown = 1.0 - occupied - empty;
    auto zero_or = [](double x, double lo, double hi) { return __double_as_longlong(x) == 0ll || (x >= lo && x <= hi); };
    const bool obs_ok = !in || (unknown >= 0.5 && unknown <= 1.0 && zero_or(empty, 0x1p-40, 1.0) && zero_or(occupied, 0x1p-40, 1.0));
    const double csum = (c.c0 + c.c1) + c.c2;
    const bool cell_ok = __double_as_longlong(c.c3) == 0ll && zero_or(c.c0, 0x1p-300, 2.0) && zero_or(c.c1, 0x1p-300, 2.0) &&
                         zero_or(c.c2, 0x1p-300, 2.0) && csum >= 0.5 && csum <= 2.0;
    if (__all(obs_ok) && cell_ok) {  // (cell_ok is wave-uniform: every lane holds the same cell state)
      double l0 = c.c0, l1 = c.c1, l2 = c.c2;
      for (; t < n_here; ++t) {
        const double qt = (kReadsQuality && a.beam_quality) ? mu_readlane(ql, t) : a.quality;
        const double pt = mu_readlane(p, t);
        const double eqt = mu_readlane(q, t) * qt;
        const double r2 = pt * eqt, r1 = (1 - pt) * eqt;
        const double r0 = 1.0 - r2 - r1;
        const double t0 = l0 * r0;
        const double t1 = (l0 * r1 + l1 * r0) + l1 * r1;
        const double t2 = (l0 * r2 + l2 * r0) + l2 * r2;
        const double t3 = l1 * r2 + l2 * r1;
        const double tot = t0 + t1 + t2 + t3;
        const double rt = mu_refined_rcp(tot);
        const double n0 = mu_div_nofix(t0, tot, rt), n1 = mu_div_nofix(t1, tot, rt), n2 = mu_div_nofix(t2, tot, rt);
        const double weight = n0 + n1 + n2;
        const double rw = mu_refined_rcp(weight);
        l0 = mu_div_nofix(n0, weight, rw);
        l1 = mu_div_nofix(n1, weight, rw);
        l2 = mu_div_nofix(n2, weight, rw);
      }
      c.c0 = l0;
      c.c1 = l1;
      c.c2 = l2;
      c.c3 = 0.0;
      return;
    }
  }
  while (t < n_here) {
    // (every lane holds the same cell state; the first lane's test keeps `t` wave-uniform)
    if (RULE == 4 && __builtin_amdgcn_readfirstlane((int)(c.c0 == 0.0))) {  // skip the free run ahead
      const unsigned long long ahead = busy >> t;
      const int stop = ahead ? min(n_here, t + __ffsll((long long)ahead) - 1) : n_here;
      if (stop > t) {
        c.x1 += (double)(stop - t);
        c.c0 = 0.0;
        t = stop;
        continue;
      }
    }
    const double qt = (kReadsQuality && a.beam_quality) ? mu_readlane(ql, t) : a.quality;
    mu_step<RULE>(qt, c, mu_readlane(p, t), RULE == 3 ? mu_readlane(q, t) : 0.0, [&](double *x, double *y) {
      *x = mu_readlane(ox, t);
      *y = mu_readlane(oy, t);
    });
    ++t;
  }
}

// Chains that run past the end of their wave (at most one per wave; the robot's own cell takes one update per
// beam, its neighbours hundreds): one thread walking such a chain pays a memory round trip per 8 records (330 us
// for 1080 updates).  Here the WAVE that holds the chain's head streams it (at the end of k_mu_apply, after the
// chains that end inside the wave): 64 records per coalesced load, then every lane applies
// them in order from broadcast values -- the same sequential arithmetic, executed redundantly by all
// lanes, so the result is bit-identical to the one-thread walk.  GMapping cells: a run of free
// observations of a cell whose mean is 0 only counts tries (see mu_step), so the run is skipped in one
// step from the ballot of the hits -- the chains around the robot are nothing but such runs.
template <typename Key, int RULE>
__device__ __forceinline__ void mu_apply_long_chains(const MuArgs &a, const Key *keys, unsigned total, unsigned i,
                                                     int lane, bool is_long) {
  constexpr Key kInvalid = ~Key(0);
  unsigned long long todo = __ballot(is_long);
  while (todo) {
    const int src = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)i, src);
    const Key hkey = keys[hi];
    const size_t at = mu_cell_index<Key>(a, hkey);
    MuCell c = mu_cell_load<RULE>(a, at);
    const MuCell was = c;
    for (unsigned j0 = hi;; j0 += 64) {
      const unsigned j = j0 + lane;
      const bool in = j < total;
      const Key k = in ? keys[j] : kInvalid;
      const double p = in ? a.rec_prob[j] : 0.0;
      const double q = (RULE == 3 && in) ? a.rec_qual[j] : 0.0;
      // this record's update quality: scan quality x its beam's (uniform branch: a kernel argument)
      constexpr bool kReadsQuality = RULE >= 1 && RULE <= 3;
      double ql = a.quality;
      if (kReadsQuality && a.beam_quality && in) ql = a.quality * a.beam_quality[a.rec_beam[j]];
      const unsigned long long m = __ballot(in && k == hkey);
      const int n_here = (m == ~0ull) ? 64 : (__ffsll((long long)~m) - 1);
      double ox = 0.0, oy = 0.0;
      if (RULE == 4 && in && !(p <= 0.5) && !isnan(p)) {  // hits: the obstacle point of the record's beam
        const unsigned b = a.rec_beam[j];
        ox = a.beam_end[2 * b];
        oy = a.beam_end[2 * b + 1];
      }
      mu_wave_apply<RULE>(a, c, lane, n_here, in, p, q, ql, ox, oy);
      if (n_here < 64) break;
    }
    if (lane == src) mu_cell_store<RULE>(a, at, c, was);
  }
}

// Chains that end inside their wave: one thread per distinct cell -- the one holding the chain's first record --
// applies its records in order.  The records of a chain sit in the lanes behind their head: every lane
// loads its own key and observation (coalesced), the heads pull them across with a lane shuffle per
// step; a chain that runs past the end of its wave goes to mu_apply_long_chains.
template <typename Key, int RULE>
__global__ __launch_bounds__(256) void k_mu_apply(MuArgs a, const Key *keys, unsigned total) {
  constexpr Key kInvalid = ~Key(0);
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool in = i < total;
  const Key key = in ? keys[i] : kInvalid;
  const double p = in ? a.rec_prob[i] : 0.0;
  const double q = (RULE == 3 && in) ? a.rec_qual[i] : 0.0;
  constexpr bool kReadsQuality = RULE >= 1 && RULE <= 3;
  double ql = a.quality;  // this record's update quality: scan quality x its beam's
  if (kReadsQuality && a.beam_quality && in) ql = a.quality * a.beam_quality[a.rec_beam[i]];
  Key prev = __shfl_up(key, 1, 64);
  if (lane == 0) prev = (in && i > 0) ? keys[i - 1] : kInvalid;
  const bool start = !in || i == 0 || prev != key;  // first record of a run of equal keys
  const unsigned long long starts = __ballot(start);
  bool head = start && key != kInvalid;
  // the run ends in front of the next start; none behind this lane: it reaches the end of the wave
  const unsigned long long behind = lane == 63 ? 0ull : starts >> (lane + 1);
  const int len_here = behind ? __ffsll((long long)behind) : 64 - lane;
  const bool open_end = !behind;  // may continue in the next wave
  // a chain that runs on into the next wave (at most one per wave: its last) is streamed by the whole wave below --
  // one coalesced load per 64 records instead of the head thread's own dependent reads (8 records per round trip:
  // 7 of a single-scan update's 19 us)
  bool is_long = false;
  if (head && open_end && i + len_here < total && keys[i + len_here] == key) {
    head = false;
    is_long = true;
  }
  size_t at = 0;
  MuCell c{0, 0, 0, 0, 0, 0};
  if (head) {
    at = mu_cell_index<Key>(a, key);
    c = mu_cell_load<RULE>(a, at);
  }
  const MuCell was = c;
  for (int t = 0; __any(head && t < len_here); ++t) {
    const double pt = __shfl(p, lane + t, 64);
    const double qt = RULE == 3 ? __shfl(q, lane + t, 64) : 0.0;
    const double qlt = (kReadsQuality && a.beam_quality) ? __shfl(ql, lane + t, 64) : a.quality;
    if (head && t < len_here)
      mu_step<RULE>(qlt, c, pt, qt, [&](double *x, double *y) {
        const unsigned b = a.rec_beam[i + t];
        *x = a.beam_end[2 * b];
        *y = a.beam_end[2 * b + 1];
      });
  }
  if (head) mu_cell_store<RULE>(a, at, c, was);
  mu_apply_long_chains<Key, RULE>(a, keys, total, i, lane, is_long);
}

// Last kernel of an update on the low-latency path: the status words go to pinned host memory and the
// launch number is published where the host spins (k_publish, score_kernels.hip) -- no copy engine, no
// hipStreamSynchronize.
__global__ void k_mu_finish(const int *error_flag, const unsigned long long *n_padding,
                            unsigned long long *h_status, unsigned *flag, unsigned seq) {
  h_status[0] = (unsigned long long)*error_flag;
  h_status[1] = *n_padding;
  *const_cast<int *>(error_flag) = 0;  // the next update may start without k_mu_count (fused into k_mu_emit)
  *const_cast<unsigned long long *>(n_padding) = 0ull;  // (the gather form only adds to it)
  __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// deferred updates (mu_set_deferred): the status words of all queued updates at once, slots back to zero
__global__ void k_mu_finish_ring(int *err, unsigned long long *pad, int n, unsigned long long *h_ring, unsigned *flag,
                                 unsigned seq) {
  const int k = threadIdx.x;
  if (k < n) {
    h_ring[2 * k] = (unsigned long long)err[k];
    h_ring[2 * k + 1] = pad[k];
    err[k] = 0;
    pad[k] = 0ull;
  }
  __syncthreads();
  if (k == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// the apply kernel for the cell kind of `a.rule`
template <typename Key>
void mu_launch_apply(const MuArgs &a, const Key *keys, unsigned total, hipStream_t stream) {
  const dim3 grid((total + 255) / 256), block(256);
#define SLAMHIP_MU_RULE(R)                                                                                   \
  case R:                                                                                                    \
    hipLaunchKernelGGL((k_mu_apply<Key, R>), grid, block, 0, stream, a, keys, total);                        \
    break;
  switch (a.rule) {
    SLAMHIP_MU_RULE(0)
    SLAMHIP_MU_RULE(1)
    SLAMHIP_MU_RULE(2)
    SLAMHIP_MU_RULE(3)
    default:
      SLAMHIP_MU_RULE(4)
  }
#undef SLAMHIP_MU_RULE
}

}  // namespace slamhip
