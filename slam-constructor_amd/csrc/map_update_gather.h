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
//               the sequential walk (mu_walk_beam: tie rule, Bresenham fail-over), leaves its cells as keys and marks
//               them in a bitmap of the window
//   k_mu_cells  far cells: one thread per cell (candidates, observations, `cell += observation` in beam order);
//               the cells around the robot, which nearly every beam visits: one WAVE per cell (64 beams tested at a
//               time, observations in parallel, applied in order by mu_wave_apply); a cell marked in the bitmap asks
//               every beam, the irregular ones by looking for its key among theirs
// HBM traffic: the touched cells read and written once, 32 bytes per beam of closed form, and one bit per window cell.
#pragma once

namespace slamhip {

constexpr int kGatherLut = 4096;  // bins of the angle table over [0, 2 pi)
constexpr double kTwoPi = 6.283185307179586476925286766559;

// ---- k_mu_lines -------------------------------------------------------------------------------------------------
template <int EST>
__global__ __launch_bounds__(256) void k_mu_lines(MuArgs a) {
  __shared__ int s_ok[4];
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const MuJob j = mu_job(a, b);
  double wx, wy;
  mu_endpoint(a, j, b, &wx, &wy);
  const double ddx = wx - j.px, ddy = wy - j.py;
  unsigned cap = 0u;
  int ex = 0, ey = 0;
  const int bx = (int)floor(j.px / a.scale), by = (int)floor(j.py / a.scale);
  if (!(a.max_range_sq < ddx * ddx + ddy * ddy)) {
    const MuBeam bm = mu_beam<EST>(a, j, b, wx, wy);
    ex = bm.ex;
    ey = bm.ey;
    cap = (unsigned)(abs(ex - bx) + abs(ey - by) + 1);
    if (t == 0) a.beam_info[b] = bm;
  }
  if (t == 0) {
    a.beam_end[2 * b] = wx;
    a.beam_end[2 * b + 1] = wy;
    if (EST > 0) {
      a.beam_inv[2 * b] = 1.0 / ddx;
      a.beam_inv[2 * b + 1] = 1.0 / ddy;
    }
    a.counts[b] = cap;
    a.offsets[b] = a.host_offsets[b];
  }
  MuLine line{0.0, 0.0, 0.0, cap, 0u};
  if (cap == 0) {
    if (t == 0) a.lines[b] = line;
    return;
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
  const int steps_x = abs(ex - bx), steps_y = abs(ey - by);
  bool ok = absA > 0.0 && absB > 0.0;
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
      ok = ok && i == steps_x && jj == steps_y;
    }
  }
  ok = __all(ok);
  if (lane == 0) s_ok[wave] = ok ? 1 : 0;
  __syncthreads();
  ok = s_ok[0] && s_ok[1] && s_ok[2] && s_ok[3];
  if (t != 0) return;
  // the two ends inside the map: a monotone walk between them stays inside (the window is clipped to the map; what
  // lies outside is not updated and reported)
  const unsigned w = (unsigned)a.width, h = (unsigned)a.height;
  const bool ends_in = (unsigned)(bx + a.origin_x) < w && (unsigned)(by + a.origin_y) < h &&
                       (unsigned)(ex + a.origin_x) < w && (unsigned)(ey + a.origin_y) < h;
  line.q0 = q0;
  line.absA = absA;
  line.invW = inv_W;
  line.flags = (ok ? 1u : 0u) | (inc_x > 0 ? 2u : 0u) | (inc_y > 0 ? 4u : 0u);
  a.lines[b] = line;
  if (ok) {
    if (!ends_in) *a.error_flag = 1;
    return;
  }
  // an irregular beam: the sequential walk decides; its cells stay behind as keys (padding: ~0) and as bits
  if ((unsigned long long)a.offsets[b] + cap > a.keys_cap) {
    *a.error_flag = 2;
    a.lines[b].cap = 0u;
    return;
  }
  __threadfence();  // counts / offsets / beam_end / beam_info above, read back by the walk
  mu_walk_beam<unsigned>(a, b);
  const unsigned *keys = (const unsigned *)a.keys + a.offsets[b];
  unsigned long long pad = 0ull;
  for (unsigned k = 0; k < cap; ++k) {
    const unsigned key = keys[k];
    if (key >= a.n_bins) ++pad;
    else atomicOr(&a.irr_bits[key >> 5], 1u << (key & 31u));
  }
  if (pad) atomicAdd(a.n_padding, pad);
}

// ---- k_mu_cells -------------------------------------------------------------------------------------------------
// does beam b (closed form L) stand on the cell (dxc, dyc) away from the robot's at step k = |dxc| + |dyc| ?
__device__ __forceinline__ bool mu_line_visits(const MuLine &L, int dxc, int dyc, unsigned k) {
  if (!(L.flags & 1u) || k >= L.cap) return false;
  const bool xpos = (L.flags & 2u) != 0u, ypos = (L.flags & 4u) != 0u;
  if ((dxc > 0 && !xpos) || (dxc < 0 && xpos) || (dyc > 0 && !ypos) || (dyc < 0 && ypos)) return false;
  const double fj = floor((L.q0 + (double)k * L.absA) * L.invW);
  const int j = (int)fmin(fmax(fj, 0.0), (double)k);
  return j == abs(dyc);
}
// ... or, an irregular beam: is the cell's key among the keys its sequential walk left?
__device__ __forceinline__ bool mu_irregular_visits(const MuArgs &a, int b, unsigned key) {
  const unsigned cap = a.counts[b];
  const unsigned *keys = (const unsigned *)a.keys + a.offsets[b];
  for (unsigned k = 0; k < cap; ++k)
    if (keys[k] == key) return true;
  return false;
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
  if (!(dist_sq > half * half * 1.0001)) return MuCand{0, n - 1, 0, -1};
  const double w = asin(half / sqrt(dist_sq)) + 1e-6;
  double rel = atan2(vy, vx) - a.theta - a.rel_a0;
  rel -= kTwoPi * floor(rel / kTwoPi);  // [0, 2 pi)
  const double lo = rel - w, hi = rel + w;
  if (!(w < 1.5)) return MuCand{0, n - 1, 0, -1};
  const double inv_bin = (double)a.lut_bins / kTwoPi;
  // beams with relative angle in [x, y]: indices lut[bin(x)] .. lut[bin(y) + 1] - 1, one more on either side
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
    c = MuCand{0, last_of(hi), first_of(lo + kTwoPi), n - 1};
  } else if (hi >= kTwoPi) {
    c = MuCand{0, last_of(hi - kTwoPi), first_of(lo), n - 1};
  } else {
    c = MuCand{first_of(lo), last_of(hi), 0, -1};
  }
  if (c.hi1 >= c.lo1 && c.lo1 <= c.hi0 + 1) c = MuCand{0, n - 1, 0, -1};  // the two stretches meet
  return c;
}

// the observation of (beam b, cell) ready for mu_step: probability (NaN: dropped), TBM quality, update quality
template <int RULE, int EST>
__device__ __forceinline__ void mu_observe(const MuArgs &a, int b, int cx, int cy, double *p, double *q, double *ql) {
  double2 pq = mu_value<EST>(a, b, cx, cy, a.beam_info + b);
  if (RULE != 3 && RULE != 0 && isnan(pq.y)) pq.x = pq.y;
  *p = pq.x;
  *q = pq.y;
  *ql = (RULE >= 1 && RULE <= 3 && a.beam_quality) ? a.quality * a.beam_quality[b] : a.quality;
}

// near cells: Chebyshev distance <= near_r from the robot's cell, (2 near_r + 1)^2 of them, one wave each, placed
// behind the workgroups of the far cells
template <int RULE, int EST>
__global__ __launch_bounds__(256) void k_mu_cells(MuArgs a, unsigned far_blocks, unsigned long long *h_status,
                                                  unsigned *flag, unsigned seq) {
  __shared__ double s_buf[4][5][64];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int rbx = a.robot_ix, rby = a.robot_iy;  // robot cell, internal coordinates
  const int side = 2 * a.near_r + 1;
  if (blockIdx.x < far_blocks) {
    // ---- far cells: one thread per cell of the window, 64 x 4 cells per workgroup
    const unsigned tiles_x = ((unsigned)a.key_w + 63u) / 64u;
    const int ix = a.key_x0 + (int)(blockIdx.x % tiles_x) * 64 + lane;
    const int iy = a.key_y0 + (int)(blockIdx.x / tiles_x) * 4 + wave;
    const bool inside = ix < a.key_x0 + a.key_w && iy < a.key_y0 + a.key_h;
    const int dxc = ix - rbx, dyc = iy - rby;
    const bool near = abs(dxc) <= a.near_r && abs(dyc) <= a.near_r;
    const unsigned key = inside ? (unsigned)(iy - a.key_y0) * (unsigned)a.key_w + (unsigned)(ix - a.key_x0) : 0u;
    // the irregular bitmap: one word per 32 cells of a row-major window -- read by every cell of it, cleared by the
    // cell that finds its bit set
    const bool irregular = inside && ((a.irr_bits[key >> 5] >> (key & 31u)) & 1u) != 0u;
    if (inside && !near) {
      const int cx = ix - a.origin_x, cy = iy - a.origin_y;  // external cell
      const unsigned k = (unsigned)(abs(dxc) + abs(dyc));
      MuCand cand;
      if (irregular) {
        cand = MuCand{0, a.n - 1, 0, -1};
      } else {
        const double vx = (cx + 0.5) * a.scale - a.px, vy = (cy + 0.5) * a.scale - a.py;
        cand = mu_candidates(a, vx, vy);
      }
      const size_t at = (size_t)iy * a.pitch + ix;
      MuCell c{0, 0, 0, 0, 0, 0};
      MuCell was = c;
      bool loaded = false;
      for (int part = 0; part < 2; ++part) {
        const int lo = part ? cand.lo1 : cand.lo0, hi = part ? cand.hi1 : cand.hi0;
        for (int b = lo; b <= hi; ++b) {
          const MuLine L = a.lines[b];
          bool visits = mu_line_visits(L, dxc, dyc, k);
          if (irregular && !(L.flags & 1u) && L.cap) visits = mu_irregular_visits(a, b, key);
          if (!visits) continue;
          if (!loaded) {
            c = mu_cell_load<RULE>(a, at);
            was = c;
            loaded = true;
          }
          double p, q, ql;
          mu_observe<RULE, EST>(a, b, cx, cy, &p, &q, &ql);
          mu_step<RULE>(ql, c, p, q, [&](double *x, double *y) {
            *x = a.beam_end[2 * b];
            *y = a.beam_end[2 * b + 1];
          });
        }
      }
      if (loaded) mu_cell_store<RULE>(a, at, c, was);
    }
    if (irregular && !near) atomicAnd(&a.irr_bits[key >> 5], ~(1u << (key & 31u)));
  } else {
    // ---- near cells: one wave per cell, 64 beams at a time
    const unsigned cell_id = (blockIdx.x - far_blocks) * 4u + (unsigned)wave;
    if (cell_id < (unsigned)(side * side)) {
      const int dxc = (int)(cell_id % (unsigned)side) - a.near_r, dyc = (int)(cell_id / (unsigned)side) - a.near_r;
      const int ix = rbx + dxc, iy = rby + dyc;
      const bool inside = ix >= a.key_x0 && ix < a.key_x0 + a.key_w && iy >= a.key_y0 && iy < a.key_y0 + a.key_h;
      if (inside) {  // (wave-uniform)
        const int cx = ix - a.origin_x, cy = iy - a.origin_y;
        const unsigned k = (unsigned)(abs(dxc) + abs(dyc));
        const unsigned key = (unsigned)(iy - a.key_y0) * (unsigned)a.key_w + (unsigned)(ix - a.key_x0);
        const bool irregular = ((a.irr_bits[key >> 5] >> (key & 31u)) & 1u) != 0u;
        MuCand cand;
        if (irregular) {
          cand = MuCand{0, a.n - 1, 0, -1};
        } else {
          const double vx = (cx + 0.5) * a.scale - a.px, vy = (cy + 0.5) * a.scale - a.py;
          cand = mu_candidates(a, vx, vy);
        }
        const size_t at = (size_t)iy * a.pitch + ix;
        MuCell c = mu_cell_load<RULE>(a, at);
        const MuCell was = c;
        bool any = false;
        for (int part = 0; part < 2; ++part) {
          const int lo = part ? cand.lo1 : cand.lo0, hi = part ? cand.hi1 : cand.hi0;
          for (int b0 = lo; b0 <= hi; b0 += 64) {
            const int b = b0 + lane;
            bool visits = false;
            if (b <= hi) {
              const MuLine L = a.lines[b];
              visits = mu_line_visits(L, dxc, dyc, k);
              if (irregular && !(L.flags & 1u) && L.cap) visits = mu_irregular_visits(a, b, key);
            }
            double p = 0.0, q = 0.0, ql = a.quality, ox = 0.0, oy = 0.0;
            if (visits) {
              mu_observe<RULE, EST>(a, b, cx, cy, &p, &q, &ql);
              if (RULE == 4 && !(p <= 0.5) && !isnan(p)) {
                ox = a.beam_end[2 * b];
                oy = a.beam_end[2 * b + 1];
              }
            }
            const unsigned long long m = __ballot(visits);
            if (!m) continue;
            any = true;
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
            mu_wave_apply<RULE>(a, c, lane, n_here, in, pc, qc, qlc, oxc, oyc);
          }
        }
        if (any && lane == 0) mu_cell_store<RULE>(a, at, c, was);
        if (irregular && lane == 0) atomicAnd(&a.irr_bits[key >> 5], ~(1u << (key & 31u)));
      }
    }
  }
  // ---- the last workgroup through hands the update's status words to the host (awaited updates only)
  if (!h_status) return;
  __shared__ unsigned s_last;
  __threadfence();
  __syncthreads();
  if (t == 0) s_last = atomicAdd(a.done_count, 1u) + 1u == gridDim.x ? 1u : 0u;
  __syncthreads();
  if (!s_last || t != 0) return;
  *a.done_count = 0u;
  h_status[0] = (unsigned long long)__hip_atomic_load(a.error_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  h_status[1] = __hip_atomic_load(a.n_padding, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  *a.error_flag = 0;
  *a.n_padding = 0ull;
  __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace slamhip
