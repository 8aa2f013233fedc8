// map_update_gather.h -- K6 for ONE scan as a GATHER: every cell of the key window collects its own observations.
//
// GridMapScanAdder::append_scan -> WallDistanceBlurringScanAdder::handle_scan_point (grid_map_scan_adders.h:54-75,
// 138-172) is a scatter on the CPU: beam after beam walks its cells (RegularSquaresGrid::world_to_cells,
// regular_squares_grid.h:56-101) and updates each one.  What a cell ends up with depends on the ORDER its observations
// arrive in -- beam order -- which is why the first K6 emitted (cell, beam) records and sorted them (radix sort,
// then a counting sort over the window: 9 dispatches, 31 MB of HBM traffic for 10.7 MB of algorithmic bytes).
// But the walk of a beam has a closed form (k_mu_emit): step k of beam b stands i_k(b) steps in x and j_k(b) steps in y
// from the robot's cell, j_k = clamp(floor((q0 + k |A|) / (|A| + |B|)), 0, k).  So a CELL can ask: it lies
// k = |dx| + |dy| steps from the robot's cell, and beam b visits it iff b's walk is at least k + 1 cells long, points
// into the cell's quadrant and has j_k(b) = |dy|.  Which beams to ask: the walk keeps the centre of every cell it
// visits within (|A| + |B|) / 2 |d| <= 0.7072 cell sides of the beam's line (its error term IS that distance, and each
// step takes the smaller of |e + A|, |e + B|, which differ by |A| + |B|), so only beams whose direction lies within
// asin(0.75 side / distance) of the direction of the cell's centre can visit it -- a handful, found through a table
// over the scan's (ascending) beam angles.  The cell then applies what it found in ascending beam order: the
// reference's order, with no record ever written.  Two kernels per update:
//   k_mu_lines  one workgroup per beam: end point, the beam's observation constants (MuBeam), the closed form
//               checked against the recurrence step by step exactly like k_mu_emit does; a beam that fails the check
//               (ties along diagonals, axis-parallel beams, walks rounding sends astray) is IRREGULAR: thread 0 runs
//               the sequential walk (mu_walk_beam: tie rule, Bresenham fail-over), leaves its cells as keys and writes
//               its number into their words of a marker array over the window (two or three beams of a scan end in
//               a cell the closed form does not reach: the reference's Bresenham fail-over is an everyday event)
//   k_mu_cells  far cells: one thread per cell (candidates, observations, `cell += observation` in beam order);
//               the cells around the robot, which nearly every beam visits: one WAVE per cell (64 beams tested at a
//               time, observations in parallel, applied in order by mu_wave_apply); a cell an irregular beam visits
//               finds that beam's number in its word of the window's marker array and takes its observation in place
// HBM traffic: the touched cells read and written once, 64 bytes per beam of closed form, and one 4-byte marker per
// window cell (read by every cell, written by the few an irregular beam visits).
#pragma once

namespace slamhip {

constexpr int kGatherLut = 4096;  // bins of the angle table over [0, 2 pi)
constexpr double kTwoPi = 6.283185307179586476925286766559;

// ---- k_mu_lines -------------------------------------------------------------------------------------------------
template <int EST>
__global__ __launch_bounds__(256) void k_mu_lines(MuArgs a) {
  __shared__ int s_ok[4];
  __shared__ double s_c[8];   // e0, A, B, q0, absA, inv_W, absB of the beam (made by wave 0 alone)
  __shared__ int s_i[8];
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  MuLine line{0.0, 0.0, 0.0, 0u, 0u, 0, 0, 0.0, 0.0, 0.0};  // (thread 0 keeps it and writes it once, at the end)
  if (wave == 0) {
    const MuJob j = mu_job(a, b);
    double wx, wy;
    mu_endpoint(a, j, b, &wx, &wy);
    const double ddx = wx - j.px, ddy = wy - j.py;
    unsigned cap = 0u;
    int ex = 0, ey = 0;
    const int bx = (int)floor(j.px / a.scale), by = (int)floor(j.py / a.scale);
    MuBeam bm{0, 0, 0.0, 0.0, 0.0, 0.0};
    if (!(a.max_range_sq < ddx * ddx + ddy * ddy)) {
      bm = mu_beam<EST>(a, j, b, wx, wy);
      ex = bm.ex;
      ey = bm.ey;
      cap = (unsigned)(abs(ex - bx) + abs(ey - by) + 1);
      if (t == 0) a.beam_info[b] = bm;
    }
    // (k_mu_emit's closed form, the same expressions)
    const double scale = a.scale;
    const int inc_x = 0 < ddx ? 1 : -1, inc_y = 0 < ddy ? 1 : -1;
    const double mid_x = (bx + 0.5) * scale, mid_y = (by + 0.5) * scale;
    const double mid_cell_seg_y = ddx * j.py + (mid_x - j.px) * ddy;
    const double e0 = mid_cell_seg_y - mid_y * ddx;
    const double A = inc_x * scale * ddy, B = -inc_y * scale * ddx;
    const double absA = fabs(A), absB = fabs(B), W = absA + absB;
    const double sgn = A < 0 ? -1.0 : 1.0;
    const double theta = (absB - absA) * 0.5;
    const double q0 = sgn * e0 - theta + absB;
    const double inv_W = 1.0 / W;
    if (t == 0) {
      a.beam_end[2 * b] = wx;
      a.beam_end[2 * b + 1] = wy;
      if (EST > 0) {
        a.beam_inv[2 * b] = 1.0 / ddx;
        a.beam_inv[2 * b + 1] = 1.0 / ddy;
      }
      a.counts[b] = cap;
      if (cap)
        line = MuLine{q0, absA, inv_W, cap, (inc_x > 0 ? 2u : 0u) | (inc_y > 0 ? 4u : 0u), ex, ey, bm.base_prob,
                      bm.base_qual, bm.hole_dist_sq};
      s_c[0] = e0;
      s_c[1] = A;
      s_c[2] = B;
      s_c[3] = q0;
      s_c[4] = absA;
      s_c[5] = inv_W;
      s_c[6] = absB;
      s_i[0] = (int)cap;
      s_i[1] = ex - bx;
      s_i[2] = ey - by;
      // the two ends inside the map: a monotone walk between them stays inside (the window is clipped to the map;
      // what lies outside is not updated and reported)
      const unsigned w = (unsigned)a.width, h = (unsigned)a.height;
      s_i[3] = ((unsigned)(bx + a.origin_x) < w && (unsigned)(by + a.origin_y) < h &&
                (unsigned)(ex + a.origin_x) < w && (unsigned)(ey + a.origin_y) < h) ? 1 : 0;
      s_i[4] = bx;
      s_i[5] = by;
    }
  }
  __syncthreads();
  const unsigned cap = (unsigned)s_i[0];
  if (cap == 0) {
    if (t == 0) {
      a.lines[b] = line;
      a.bad[b] = 0;
    }
    return;
  }
  const double e0 = s_c[0], A = s_c[1], B = s_c[2], q0 = s_c[3], absA = s_c[4], inv_W = s_c[5], absB = s_c[6];
  const int dxx = s_i[1], dyy = s_i[2];
  const int steps_x = abs(dxx), steps_y = abs(dyy);
  // every step's decision checked against the recurrence (as in k_mu_emit); whether the walk ENDS on the end cell is
  // kept apart: a walk whose steps all check out but which stands on another cell after its cap cells is the
  // reference's everyday fail-over to Bresenham -- two or three beams of a scan
  bool ok = absA + absB > 0.0, end_ok = true;  // (axis-parallel beams included: the check decides)
  for (unsigned k = (unsigned)t; k < cap && ok; k += 256u) {
    const double fj = floor((q0 + (double)k * absA) * inv_W), fjn = floor((q0 + (double)(k + 1) * absA) * inv_W);
    const int jj = (int)fmin(fmax(fj, 0.0), (double)k), jn = (int)fmin(fmax(fjn, 0.0), (double)(k + 1));
    const int i = (int)k - jj;
    const double e = e0 + (double)i * A + (double)jj * B;
    const double d = fabs(e + B) - fabs(e + A);
    if (k + 1 < cap) {
      const bool x_formula = jn == jj;  // the formula's next step is an x step
      ok = ok && fabs(d) > 2e-7 && (0 < d) == x_formula && jn - jj <= 1;
    } else {
      end_ok = i == steps_x && jj == steps_y;
    }
  }
  ok = __all(ok);
  end_ok = __all(end_ok);
  if (lane == 0) s_ok[wave] = (ok ? 1 : 0) | (end_ok ? 2 : 0);
  __syncthreads();
  const int all = s_ok[0] & s_ok[1] & s_ok[2] & s_ok[3];
  ok = (all & 1) != 0;
  end_ok = (all & 2) != 0;
  if (ok && end_ok) {
    if (t == 0) {
      line.flags |= 1u;
      a.lines[b] = line;
      a.bad[b] = 0;
      if (!s_i[3]) *a.error_flag = 1;
    }
    return;
  }
  // ---- an irregular beam: its cells stay behind as keys (padding: ~0) and as marker words
  if (t == 0) {
    a.lines[b] = line;  // (flags without the ok bit)
    a.bad[b] = 1;
  }
  const unsigned base = a.host_offsets[b];  // its stretch of the key buffer, sized by the host (pinned: read here only)
  if ((unsigned long long)base + cap > a.keys_cap) {
    if (t == 0) {
      *a.error_flag = 2;
      a.lines[b].cap = 0u;
    }
    return;
  }
  if (t == 0) a.offsets[b] = base;
  unsigned *keys = (unsigned *)a.keys + base;
  auto mark = [&](unsigned key) {
    // the cell learns WHICH beam it is irregular for (beam + 1); a second irregular beam through the same cell leaves
    // the top bit: that cell looks its beams up among the keys
    const unsigned old = atomicCAS(&a.irr_bits[key], 0u, (unsigned)b + 1u);
    if (old != 0u && (old & 0x7fffffffu) != (unsigned)b + 1u) atomicOr(&a.irr_bits[key], 0x80000000u);
  };
  if (ok) {
    // Steps fine, end missed: the recurrence emits its cap cells, none of them the end cell (only the last one has its
    // L1 distance), steps once more -- further away still -- and gives up: DiscreteSegment2D's Bresenham
    // (geometry_discrete_primitives.h:55-104) from the robot's cell to the end cell replaces the walk.  Its error
    // recurrence picks, at primary step i, the secondary count s minimising |i S - s P| with ties to the larger s
    // (`abs(err_inc_primary) < abs(err_inc_both)` is strict): s_i = floor((2 i S + P) / 2 P) -- integers, so every
    // thread makes its own cell (thread 0 alone took 12 us for it: the kernel's whole duration).
    const bool y_is_primary = abs(dxx) < abs(dyy);
    const int d_primary = y_is_primary ? dyy : dxx, d_secondary = y_is_primary ? dxx : dyy;
    const long long P = llabs((long long)d_primary), S = llabs((long long)d_secondary);
    const int inc_primary = 0 < d_primary ? 1 : -1, inc_secondary = 0 < d_secondary ? 1 : -1;
    const int bx = s_i[4], by = s_i[5];
    const unsigned n_cells = (unsigned)P + 1u;
    const unsigned w = (unsigned)a.width, h = (unsigned)a.height;
    bool oob_any = false;
    for (unsigned i = (unsigned)t; i < cap; i += 256u) {
      unsigned key = ~0u;
      if (i < n_cells) {
        const long long sec = P ? (2ll * (long long)i * S + P) / (2ll * P) : 0ll;
        const int primary = (y_is_primary ? by : bx) + inc_primary * (int)i;
        const int secondary = (y_is_primary ? bx : by) + inc_secondary * (int)sec;
        const int cx = y_is_primary ? secondary : primary, cy = y_is_primary ? primary : secondary;
        const unsigned ix = (unsigned)(cx + a.origin_x), iy = (unsigned)(cy + a.origin_y);
        const bool oob = ix >= w || iy >= h;
        oob_any |= oob;
        if (!oob) key = (iy - (unsigned)a.key_y0) * (unsigned)a.key_w + (ix - (unsigned)a.key_x0);
      }
      keys[i] = key;
      if (key < a.n_bins) mark(key);
    }
    if (__any(oob_any) && lane == 0) *a.error_flag = 1;
    if (t == 0 && cap > n_cells) atomicAdd(a.n_padding, (unsigned long long)(cap - n_cells));
    // (cells outside the map are padding too, counted like the sequential walk counts them)
    unsigned long long oob_cnt = 0;
    for (unsigned i = (unsigned)t; i < n_cells; i += 256u) oob_cnt += keys[i] == ~0u ? 1ull : 0ull;
    if (oob_cnt) atomicAdd(a.n_padding, oob_cnt);
    return;
  }
  // ties: one scan in five has such a beam (a ray within 1e-7 of a grid vertex).  Wave 0 walks it piece by piece
  // (mu_walk_beam_wave; thread 0 step by step where that gives up); the marks (an atomic with a returned value each)
  // are everybody's: one thread walking and then marking 600 cells one round trip after the other was 80-200 us, the
  // kernel's whole tail.
  __threadfence();  // counts / offsets / beam_end / beam_info above (thread 0's stores), read back by the walk
  __syncthreads();
  if (wave == 0) {
    if (!mu_walk_beam_wave<unsigned>(a, b, lane) && lane == 0) mu_walk_beam<unsigned>(a, b);
    __threadfence();  // its keys, read back by the workgroup
  }
  __syncthreads();
  unsigned pad = 0u;
  for (unsigned k = (unsigned)t; k < cap; k += 256u) {
    const unsigned key = __builtin_nontemporal_load(&keys[k]);
    if (key >= a.n_bins) ++pad;
    else mark(key);
  }
  if (pad) atomicAdd(a.n_padding, (unsigned long long)pad);
}

// ---- k_mu_cells -------------------------------------------------------------------------------------------------
// does beam b (closed form L) stand on the cell (dxc, dyc) away from the robot's at step k = |dxc| + |dyc| ?
__device__ __forceinline__ bool mu_line_visits(const MuLine &L, int dxc, int dyc, unsigned k) {
  if (!(L.flags & 1u) || k >= L.cap) return false;
  const bool xpos = (L.flags & 2u) != 0u, ypos = (L.flags & 4u) != 0u;
  if ((dxc > 0 && !xpos) || (dxc < 0 && xpos) || (dyc > 0 && !ypos) || (dyc < 0 && ypos)) return false;
  // j_k = clamp(floor(t), 0, k) == |dyc|, without the floor: |dyc| <= t < |dyc| + 1, the clamps opening the interval
  // at j = 0 below and at j = k above (the same t as k_mu_emit's: one multiply-add and one multiply)
  const double t = (L.q0 + (double)k * L.absA) * L.invW;
  const unsigned jd = (unsigned)abs(dyc);
  return (jd == 0u || t >= (double)jd) && (jd == k || t < (double)(jd + 1u));
}
// ... or, an irregular beam: is the cell's key among the keys its sequential walk left?  Every step of that walk
// moves one cell in x, in y or in both (the tie rule's diagonal, Bresenham's), so a cell (dxc, dyc) away from the
// robot's can only stand at the steps max(|dxc|, |dyc|) .. |dxc| + |dyc|.
__device__ __forceinline__ bool mu_irregular_visits(const MuArgs &a, int b, unsigned key, int dxc, int dyc) {
  const unsigned cap = a.counts[b];
  const unsigned *keys = (const unsigned *)a.keys + a.offsets[b];
  const unsigned k_lo = (unsigned)max(abs(dxc), abs(dyc)), k_hi = min(cap, (unsigned)(abs(dxc) + abs(dyc)) + 1u);
  for (unsigned k = k_lo; k < k_hi; ++k)
    if (keys[k] == key) return true;
  return false;
}
// the first irregular beam at or behind `from` (INT_MAX: none); a.bad is padded with zeros to a multiple of 8 bytes
__device__ __forceinline__ int mu_next_bad(const MuArgs &a, int from) {
  const int n8 = (a.n + 7) & ~7;
  for (int b0 = from & ~7; b0 < n8; b0 += 8) {
    unsigned long long w = *reinterpret_cast<const unsigned long long *>(a.bad + b0);
    if (b0 < from) w &= ~0ull << (8 * (from - b0));  // (the bytes in front of `from`)
    if (w) {
      const int b = b0 + (__ffsll((long long)w) - 1) / 8;
      return b < a.n ? b : INT_MAX;  // (bytes behind the scan's last beam may be left from a longer scan)
    }
  }
  return INT_MAX;
}

// A direction as a PSEUDO-ANGLE in [0, 4): t = |y| / (|x| + |y|) unfolded over the four quadrants -- monotone in the
// angle, slope between 1/2 and 1 per radian, and one reciprocal instead of an atan2 (which, with the asin of the
// window's half width, was 36 of the cell kernel's first 54 us).  The host builds the beam table over the same function
// (mu_pseudo_angle_host), so only monotonicity matters, not the values.
__device__ __forceinline__ double mu_pseudo_angle(double x, double y) {
  const double ax = fabs(x), ay = fabs(y);
  const double t = ay * __builtin_amdgcn_rcp(ax + ay);  // (ax + ay > 0: the caller's vector is not zero)
  return y >= 0.0 ? (x >= 0.0 ? t : 2.0 - t) : (x < 0.0 ? 2.0 + t : 4.0 - t);
}

// the beams that can visit a cell whose centre lies (vx, vy) from the robot: index ranges [lo0, hi0] and [lo1, hi1]
// (the second one empty unless the window of directions wraps around the ends of the scan), ascending
struct MuCand {
  int lo0, hi0, lo1, hi1;
};
__device__ __forceinline__ MuCand mu_candidates(const MuArgs &a, double vx, double vy) {
  const int n = a.n;
  const double half = 0.75 * a.scale;  // (0.7072 is the bound; the rest is margin)
  const double dist_sq = vx * vx + vy * vy;
  // closer than two window half widths: every beam (those cells are near cells anyway)
  if (!(dist_sq > 4.0 * half * half)) return MuCand{0, n - 1, 0, -1};
  // the cell's direction relative to beam 0
  const double rx = a.rot_c * vx + a.rot_s * vy, ry = a.rot_c * vy - a.rot_s * vx;
  const double rel = mu_pseudo_angle(rx, ry);
  // half width of the window in radians: asin(x) <= 1.05 x for x <= 1/2 (the hardware's reciprocal square root is
  // good to 1e-6 relative); in pseudo-angle: times its slope r^2 / (|x| + |y|)^2 (between 1/2 and 1) at this
  // direction, which changes by less than a factor 1 + 3 w across a window of half width w <= 1/2
  const double w_rad = 1.051 * half * __builtin_amdgcn_rsq(dist_sq);
  const double l1 = fabs(rx) + fabs(ry);
  const double slope = fmin(1.0, dist_sq * __builtin_amdgcn_rcp(l1 * l1) * (1.0 + 3.0 * w_rad) * 1.00001);
  const double w = w_rad * slope + 1e-5;
  const double lo = rel - w, hi = rel + w;
  const double inv_bin = (double)a.lut_bins * 0.25;
  // beams with pseudo-angle in [x, y]: indices lut[bin(x)] .. lut[bin(y) + 1] - 1, one more on either side
  auto first_of = [&](double x) {
    const int m = min(max((int)floor(x * inv_bin), 0), a.lut_bins - 1);
    return max((int)a.lut[m] - 1, 0);
  };
  auto last_of = [&](double y) {
    const int m = min(max((int)floor(y * inv_bin) + 1, 0), a.lut_bins);
    return min((int)a.lut[m], n - 1);
  };
  MuCand c;
  if (lo < 0.0) {
    c = MuCand{0, last_of(hi), first_of(lo + 4.0), n - 1};
  } else if (hi >= 4.0) {
    c = MuCand{0, last_of(hi - 4.0), first_of(lo), n - 1};
  } else {
    c = MuCand{first_of(lo), last_of(hi), 0, -1};
  }
  if (c.hi1 >= c.lo1 && c.lo1 <= c.hi0 + 1) c = MuCand{0, n - 1, 0, -1};  // the two stretches meet
  return c;
}

// the observation of (beam b, cell) ready for mu_step: probability (NaN: dropped), TBM quality, update quality
template <int RULE, int EST>
__device__ __forceinline__ void mu_observe(const MuArgs &a, const MuLine &L, int b, int cx, int cy, int rcx, int rcy,
                                           double *p, double *q, double *ql) {
  MuBeam bm;
  bm.ex = L.ex;
  bm.ey = L.ey;
  bm.base_prob = L.base_prob;
  bm.base_qual = L.base_qual;
  bm.hole_dist_sq = L.hole_dist_sq;
  const double odx = rcx - L.ex, ody = rcy - L.ey;  // (mu_beam's expression: exact small integers)
  bm.obst_dist_sq = odx * odx + ody * ody;
  double2 pq = mu_value<EST>(a, b, cx, cy, &bm);
  if (RULE != 3 && RULE != 0 && isnan(pq.y)) pq.x = pq.y;
  *p = pq.x;
  *q = pq.y;
  *ql = (RULE >= 1 && RULE <= 3 && a.beam_quality) ? a.quality * a.beam_quality[b] : a.quality;
}

constexpr int kGatherChunk = 256;  // closed forms staged in LDS at a time (16 KB)

// min / max over the workgroup (256 threads), through LDS
__device__ __forceinline__ void mu_block_min_max(int *s_red, int t, int lo, int hi, int *out_lo, int *out_hi) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    lo = min(lo, __shfl_xor(lo, off, 64));
    hi = max(hi, __shfl_xor(hi, off, 64));
  }
  __syncthreads();
  if ((t & 63) == 0) {
    s_red[2 * (t >> 6)] = lo;
    s_red[2 * (t >> 6) + 1] = hi;
  }
  __syncthreads();
  *out_lo = min(min(s_red[0], s_red[2]), min(s_red[4], s_red[6]));
  *out_hi = max(max(s_red[1], s_red[3]), max(s_red[5], s_red[7]));
}

// far cells: 16 x 16 cells per workgroup, one thread each; near cells (Chebyshev distance <= near_r from the robot's
// cell, (2 near_r + 1)^2 of them): one wave each, four per workgroup, behind the far workgroups.  Either way the
// workgroup stages the closed forms of the beams its cells may ask -- the union of their candidate ranges, 512 at a
// time -- in LDS: a thread's candidates are then an LDS read apart, not an L2 round trip.
template <int RULE, int EST>
__global__ __launch_bounds__(256) void k_mu_cells(MuArgs a, unsigned far_blocks) {
  __shared__ MuLine s_line[kGatherChunk];
  __shared__ double s_buf[4][5][64];
  __shared__ int s_red[8];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int rbx = a.robot_ix, rby = a.robot_iy;  // robot cell, internal coordinates
  const int rcx = rbx - a.origin_x, rcy = rby - a.origin_y;  // ... external
  const int side = 2 * a.near_r + 1;
  // (the near workgroups come FIRST: behind 700 far ones they started when those were through, 21 us late)
  const unsigned near_blocks = gridDim.x - far_blocks;
  const bool far_wg = blockIdx.x >= near_blocks;
  const unsigned far_id = blockIdx.x - near_blocks;
  // ---- this thread's (far) or this wave's (near) cell
  int ix, iy;
  bool inside, mine;  // inside the window; a cell this thread / wave works on
  if (far_wg) {
    const unsigned tiles_x = ((unsigned)a.key_w + 15u) / 16u;
    ix = a.key_x0 + (int)(far_id % tiles_x) * 16 + (t & 15);
    iy = a.key_y0 + (int)(far_id / tiles_x) * 16 + (t >> 4);
    inside = ix < a.key_x0 + a.key_w && iy < a.key_y0 + a.key_h;
    mine = inside && !(abs(ix - rbx) <= a.near_r && abs(iy - rby) <= a.near_r);
  } else {
    const unsigned cell_id = blockIdx.x * 4u + (unsigned)wave;
    const bool exists = cell_id < (unsigned)(side * side);
    ix = rbx + (exists ? (int)(cell_id % (unsigned)side) - a.near_r : 0);
    iy = rby + (exists ? (int)(cell_id / (unsigned)side) - a.near_r : 0);
    inside = exists && ix >= a.key_x0 && ix < a.key_x0 + a.key_w && iy >= a.key_y0 && iy < a.key_y0 + a.key_h;
    mine = inside;
  }
  const int dxc = ix - rbx, dyc = iy - rby;
  const int cx = ix - a.origin_x, cy = iy - a.origin_y;  // external cell
  const unsigned k = (unsigned)(abs(dxc) + abs(dyc));
  const unsigned key = inside ? (unsigned)(iy - a.key_y0) * (unsigned)a.key_w + (unsigned)(ix - a.key_x0) : 0u;
  // the irregular word of the cell: 0, or beam + 1 of the irregular beam that visits it (top bit: more than one);
  // cleared by the cell that finds it set
  const unsigned irr_word = mine ? a.irr_bits[key] : 0u;
  const bool irregular = irr_word != 0u;
  const bool irr_many = (irr_word & 0x80000000u) != 0u;
  const int irr_beam = (int)(irr_word & 0x7fffffffu) - 1;
  MuCand cand{0, -1, 0, -1};
  if (mine) {
    if (irregular && !far_wg) {
      cand = MuCand{0, a.n - 1, 0, -1};  // (a near cell asks every beam, 64 at a time)
    } else {
      const double vx = (cx + 0.5) * a.scale - a.px, vy = (cy + 0.5) * a.scale - a.py;
      cand = mu_candidates(a, vx, vy);
    }
  }
  // a far cell visited by SEVERAL irregular beams (rare: they fan out from the robot) goes its own way below: the
  // beams of its window AND the irregular beams, in ascending order, closed forms read from memory.  With one irregular
  // visitor -- the word names it -- the cell takes that beam's observation when the loop over its window passes it.
  const MuCand own = cand;
  if (far_wg && irr_many) cand = MuCand{0, -1, 0, -1};
  bool irr_pending = far_wg && irregular && !irr_many;
  const size_t at = (size_t)iy * a.pitch + ix;
  MuCell c{0, 0, 0, 0, 0, 0};
  MuCell was = c;
  bool loaded = false;
  if (!far_wg && mine) {  // (a near cell: the whole wave holds its state)
    c = mu_cell_load<RULE>(a, at);
    was = c;
  }
  // (far cells) the observation beam b makes of this cell, applied to it
  auto take = [&](const MuLine &L, int b) {
    if (!loaded) {
      c = mu_cell_load<RULE>(a, at);
      was = c;
      loaded = true;
    }
    double p, q, ql;
    mu_observe<RULE, EST>(a, L, b, cx, cy, rcx, rcy, &p, &q, &ql);
    mu_step<RULE>(ql, c, p, q, [&](double *x, double *y) {
      *x = a.beam_end[2 * b];
      *y = a.beam_end[2 * b + 1];
    });
  };
  for (int part = 0; part < 2; ++part) {
    const int lo = part ? cand.lo1 : cand.lo0, hi = part ? cand.hi1 : cand.hi0;
    int u_lo, u_hi;
    mu_block_min_max(s_red, t, hi >= lo ? lo : INT_MAX, hi >= lo ? hi : INT_MIN, &u_lo, &u_hi);
    for (int c0 = u_lo; c0 <= u_hi; c0 += kGatherChunk) {  // (workgroup-uniform)
      const int c1 = min(u_hi, c0 + kGatherChunk - 1);
      __syncthreads();
      for (int q = t; q < (c1 - c0 + 1) * 4; q += 256)  // 16 bytes per thread and turn: coalesced
        reinterpret_cast<double2 *>(s_line)[q] = reinterpret_cast<const double2 *>(a.lines + c0)[q];
      __syncthreads();
      const int b_lo = max(lo, c0), b_hi = min(hi, c1);
      if (far_wg) {
        for (int b = b_lo; b <= b_hi; ++b) {
          if (irr_pending && b >= irr_beam) {  // the one irregular visitor, in its place among the regular ones
            take(a.lines[irr_beam], irr_beam);
            irr_pending = false;
            if (b == irr_beam) continue;
          }
          const MuLine &L = s_line[b - c0];
          if (mu_line_visits(L, dxc, dyc, k)) take(L, b);
        }
      } else if (mine) {  // (wave-uniform)
        for (int b0 = b_lo; b0 <= b_hi; b0 += 64) {
          const int b = b0 + lane;
          bool visits = false;
          double p = 0.0, q = 0.0, ql = a.quality, ox = 0.0, oy = 0.0;
          if (b <= b_hi) {
            const MuLine &L = s_line[b - c0];
            visits = mu_line_visits(L, dxc, dyc, k);
            if (irregular && !(L.flags & 1u) && L.cap) visits = mu_irregular_visits(a, b, key, dxc, dyc);
            if (visits) {
              mu_observe<RULE, EST>(a, L, b, cx, cy, rcx, rcy, &p, &q, &ql);
              if (RULE == 4 && !(p <= 0.5) && !isnan(p)) {
                ox = a.beam_end[2 * b];
                oy = a.beam_end[2 * b + 1];
              }
            }
          }
          const unsigned long long m = __ballot(visits);
          if (!m) continue;
          loaded = true;
          const int n_here = __popcll(m);
          // the visiting lanes move to the front, order kept (rank = visiting lanes below)
          const int rank = __popcll(m & ((1ull << lane) - 1ull));
          if (visits) {
            s_buf[wave][0][rank] = p;
            s_buf[wave][1][rank] = q;
            s_buf[wave][2][rank] = ql;
            s_buf[wave][3][rank] = ox;
            s_buf[wave][4][rank] = oy;
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the wave's LDS stores before its loads below
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          const bool in = lane < n_here;
          const double pc = in ? s_buf[wave][0][lane] : 0.0, qc = in ? s_buf[wave][1][lane] : 0.0;
          const double qlc = in ? s_buf[wave][2][lane] : a.quality;
          const double oxc = in ? s_buf[wave][3][lane] : 0.0, oyc = in ? s_buf[wave][4][lane] : 0.0;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // ... and those loads before the next round's stores
          __builtin_amdgcn_wave_barrier();
          // (the wave's stretch of s_buf is free again: every lane holds its record)
          mu_wave_apply<RULE>(a, c, lane, n_here, in, pc, qc, qlc, oxc, oyc, &s_buf[wave][0][0]);
        }
      }
    }
  }
  if (irr_pending) take(a.lines[irr_beam], irr_beam);  // (behind every beam of the window)
  if (far_wg && irr_many) {
    auto irregular_beam = [&](int b) {
      const MuLine L = a.lines[b];
      if (L.cap && mu_irregular_visits(a, b, key, dxc, dyc)) take(L, b);
    };
    int nb = mu_next_bad(a, 0);
    for (int part = 0; part < 2; ++part) {
      const int lo = part ? own.lo1 : own.lo0, hi = part ? own.hi1 : own.hi0;
      for (int b = lo; b <= hi; ++b) {
        while (nb < b) {
          irregular_beam(nb);
          nb = mu_next_bad(a, nb + 1);
        }
        if (b == nb) {
          irregular_beam(b);
          nb = mu_next_bad(a, nb + 1);
          continue;
        }
        const MuLine L = a.lines[b];
        if (mu_line_visits(L, dxc, dyc, k)) take(L, b);
      }
    }
    while (nb != INT_MAX) {
      irregular_beam(nb);
      nb = mu_next_bad(a, nb + 1);
    }
  }
  if (far_wg) {
    if (loaded) mu_cell_store<RULE>(a, at, c, was);
    if (irregular) a.irr_bits[key] = 0u;
  } else if (mine && lane == 0) {
    if (loaded) mu_cell_store<RULE>(a, at, c, was);
    if (irregular) a.irr_bits[key] = 0u;
  }
}

}  // namespace slamhip
