// gm_score_device.h -- the GMapping OOPE on the device, shared by K3 (score_kernels.hip) and the
// hill-climbing chain (hc_chain.hip): one definition, so a pose scored by either gets the same bits.
//   GmappingOccupancyObservationPE         src/slams/gmapping/gmapping_occupancy_observation_pe.h:17-38
//   GmappingBaseCell::discrepancy          src/slams/gmapping/gmapping_grid_cell.h:35-38
#pragma once

#include "score_device.h"

namespace slamhip {

static constexpr int kGmBlock = 256;  // the canonical layout of run resolution and sum: thread t owns beams t + 256 k
// dynamic LDS of gm_score_pose_wide behind its int arrays: one double per thread for the helper lanes' distances
template <int NT>
constexpr size_t kGmHelperDoubles = NT >= 512 ? (size_t)NT : 0;

// ---- K3: GMapping OOPE -------------------------------------------------------------------------
// value of one endpoint: max over the (2w+1)^2 window of cells with prob_occ >= th of
// exp(-|cell.obst - endpoint|^2 / 0.05); the window order of the reference (dx outer, dy inner)
// does not matter for a max of finite values.
// `tiles`: null = dense window (m.pitch); else the tile table of the pose's own copy-on-write map
// (tile_pool.h): m.payload is then the tile pool and m.width/height the virtual extent.
// The 3 x 3 window (slam/scmtch/oope/window = 1, every shipped configuration).  A thread's time in K3 is
// its chain of dependent loads: the generic loop below pays one round trip per window cell (two with a
// tile table in front), one after the other -- 9 x KB round trips per pose, ~30 of a launch's 37 us.
// Here the nine cells are fetched with independent loads, behind at most four tile-table entries (the window's
// corners), and reduced with selects: a round trip for the tiles, one for the occupancies, one for the obstacle
// means of the full cells.
// `unk`: the prototype payload in LDS.  Taken from the kernel arguments it lived in scalar registers across
// the whole kernel, and the compiler parked two of its doubles in SCRATCH (24 bytes per lane written at
// entry and read back before the gathers: the kernel's only scratch, 1.7 MB of HBM writes per launch).
// value of a window whose smallest squared distance to a full cell's obstacle mean is best_d2 (any: there is one)
// (r06, measured and left alone: glibc's exp restated -- csrc/libm_exact.h, a 128-entry table + a degree-5 polynomial, fewer
// instructions than the device library's generic exp and the reference's bits per beam -- made the phase SLOWER: its table
// gather is one more dependent load per beam; lone chain phase A 3.56 -> 3.67 us, 100-particle step 0.450 -> 0.475 ms.
// The exact modes use it, exact_kernels.hip; the fast paths keep the device's exp.)
__device__ __forceinline__ double gm_exp(double x) { return exp(x); }
__device__ __forceinline__ double gm_value_of(double best_d2, bool any) {
  if (!any) return 0.0;
  const double similarity = gm_exp(-best_d2 / 0.05);
  const double r = 1.0 - (1.0 - similarity);
  return 0.0 < r ? r : 0.0;
}

// The 3 x 3 window of one end point in stages of independent loads, so that a caller can interleave the stages of
// two windows: tile-table entries (the window's corners) -> issue_occ: the OCCUPANCY of the nine cells, 8 bytes
// each -> issue_obst: the obstacle mean (16 bytes) of the full ones only -- one to three next to a wall, none in
// free space -> finish.  (r03: fetching the nine cells whole was 288 bytes per beam through the CU's L1.)
struct GmWindow1 {
  unsigned at9[9];  // cell index, ~0: outside the map (cell indices fit 32 bits: 2^32 cells are 128 GB of payload)
  double occ9[9], obx9[9], oby9[9];
  __device__ __forceinline__ void issue_occ(const MapView &m, const double *unk, const int *tiles, int cx, int cy) {
    const int ix0 = cx + m.origin_x, iy0 = cy + m.origin_y;
    int t00 = 0, t01 = 0, t10 = 0, t11 = 0, txl = 0, tyl = 0;
    if (tiles) {
      const int tiles_y = m.height >> kTileShift;
      txl = min(max(ix0 - 1, 0) >> kTileShift, m.pitch - 1);
      tyl = min(max(iy0 - 1, 0) >> kTileShift, tiles_y - 1);
      const int txh = min(max(ix0 + 1, 0) >> kTileShift, m.pitch - 1);
      const int tyh = min(max(iy0 + 1, 0) >> kTileShift, tiles_y - 1);
      t00 = tiles[tyl * m.pitch + txl];
      t01 = tiles[tyl * m.pitch + txh];
      t10 = tiles[tyh * m.pitch + txl];
      t11 = tiles[tyh * m.pitch + txh];
    }
    const double *pay = m.payload;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int ix = ix0 + i / 3 - 1, iy = iy0 + i % 3 - 1;  // dx outer, dy inner like the reference
      const bool inb = (unsigned)ix < (unsigned)m.width && (unsigned)iy < (unsigned)m.height;
      size_t at;
      if (tiles) {
        const bool lo_x = (ix >> kTileShift) == txl, lo_y = (iy >> kTileShift) == tyl;
        const int tile = lo_y ? (lo_x ? t00 : t01) : (lo_x ? t10 : t11);
        at = ((size_t)tile << (2 * kTileShift)) + ((size_t)(iy & kTileMask) << kTileShift) + (ix & kTileMask);
      } else {
        at = (size_t)iy * m.pitch + ix;
      }
      at9[i] = inb ? (unsigned)at : ~0u;
      occ9[i] = unk[0];
      if (inb) occ9[i] = pay[4 * at];
    }
  }
  __device__ __forceinline__ void issue_obst(const MapView &m, const double *unk, const GmParams &gp) {
    const double *pay = m.payload;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      obx9[i] = unk[1];
      oby9[i] = unk[2];
      if (!(occ9[i] < gp.fullness_th) && at9[i] != ~0u) {
        obx9[i] = pay[4 * (size_t)at9[i] + 1];
        oby9[i] = pay[4 * (size_t)at9[i] + 2];
      }
    }
  }
  __device__ __forceinline__ double finish(const GmParams &gp, double ox, double oy) const {
    double best_d2 = __builtin_inf();
    bool any = false;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const double ddx = obx9[i] - ox, ddy = oby9[i] - oy;
      const double d2 = ddx * ddx + ddy * ddy;
      const bool better = !(occ9[i] < gp.fullness_th) && d2 < best_d2;
      best_d2 = better ? d2 : best_d2;
      any |= better;
    }
    return gm_value_of(best_d2, any);
  }
};

// ... and ONE cell of such a window (cell i of the nine, the reference's order): what a helper lane fetches for a
// beam that has no thread of its own (gm_score_pose_wide).  d2(): the squared distance if the cell is full, else +inf.
struct GmWindowCell {
  unsigned at;
  double occ, obx, oby;
  __device__ __forceinline__ void issue_occ(const MapView &m, const double *unk, const int *tiles, int cx, int cy, int i) {
    const int ix = cx + m.origin_x + i / 3 - 1, iy = cy + m.origin_y + i % 3 - 1;
    const bool inb = (unsigned)ix < (unsigned)m.width && (unsigned)iy < (unsigned)m.height;
    at = ~0u;
    occ = unk[0];
    if (inb) {
      size_t a;
      if (tiles) {
        const int tile = tiles[(iy >> kTileShift) * m.pitch + (ix >> kTileShift)];  // pitch = tiles per row
        a = ((size_t)tile << (2 * kTileShift)) + ((size_t)(iy & kTileMask) << kTileShift) + (ix & kTileMask);
      } else {
        a = (size_t)iy * m.pitch + ix;
      }
      at = (unsigned)a;
      occ = m.payload[4 * a];
    }
  }
  __device__ __forceinline__ void issue_obst(const MapView &m, const double *unk, const GmParams &gp) {
    obx = unk[1];
    oby = unk[2];
    if (!(occ < gp.fullness_th) && at != ~0u) {
      obx = m.payload[4 * (size_t)at + 1];
      oby = m.payload[4 * (size_t)at + 2];
    }
  }
  __device__ __forceinline__ double d2(const GmParams &gp, double ox, double oy) const {
    const double ddx = obx - ox, ddy = oby - oy;
    const double v = ddx * ddx + ddy * ddy;
    // (a NaN distance never wins in GmWindow1::finish: `d2 < best` is false; +inf here says the same)
    return (!(occ < gp.fullness_th) && v < __builtin_inf()) ? v : __builtin_inf();
  }
};

// The same window through the NEIGHBOURHOOD MASK of its centre cell (MapView::nbr_ok, dense windows): one 4-byte load
// says which of the nine cells are full, and only those cells' obstacle means are fetched -- one to three next to a
// wall, none in free space -- instead of nine occupancies first.  The phase is bound by what a beam makes the CU
// issue (nine address computations, nine loads, nine tests), not by bytes: 0.57 -> 0.47 ms per 100-particle step.
// A centre cell on the window's rim (or outside) has no complete mask: the nine-cell form above.  The selection is
// the same minimum over the same full cells.
__device__ __forceinline__ double gm_fresh_value_w1_nbr(const MapView &m, const double *unk, const GmParams &gp, int cx,
                                                        int cy, double ox, double oy) {
  const int ix0 = cx + m.origin_x, iy0 = cy + m.origin_y;
  const bool inner = ix0 >= 1 && iy0 >= 1 && ix0 + 1 < m.width && iy0 + 1 < m.height;
  if (!inner) {
    GmWindow1 w;
    w.issue_occ(m, unk, nullptr, cx, cy);
    w.issue_obst(m, unk, gp);
    return w.finish(gp, ox, oy);
  }
  const double *pay = m.payload;
  const size_t row = 4 * (size_t)m.pitch;
  const double *c0 = pay + row * (size_t)(iy0 - 1) + 4 * (size_t)(ix0 - 1);  // cell (-1, -1)
  const unsigned m9 = reinterpret_cast<const unsigned *>(c0 + row + 4 + 3)[0];
  double obx9[9], oby9[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    obx9[i] = 0.0;
    oby9[i] = 0.0;
    if ((m9 & (1u << i)) != 0u) {
      const double *c = c0 + (i % 3) * row + 4 * (i / 3);  // dx outer, dy inner like the reference
      obx9[i] = c[1];
      oby9[i] = c[2];
    }
  }
  double best_d2 = __builtin_inf();
  bool any = false;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const double ddx = obx9[i] - ox, ddy = oby9[i] - oy;
    const double d2 = ddx * ddx + ddy * ddy;
    const bool better = (m9 & (1u << i)) != 0u && d2 < best_d2;  // (a NaN distance never wins)
    best_d2 = better ? d2 : best_d2;
    any |= better;
  }
  return gm_value_of(best_d2, any);
}

// ... and through the tiles of a pool (tile_pool.h): a tile's masks know the cells of their own tile only -- a tile is
// shared by maps whose neighbouring tiles differ.  A centre cell on its tile's rim (3 % of them) asks the cells across
// the rim for THEIR masks: the one next to it in x gives the three cells of that column, the one in y the three of
// that row, the one across the corner itself -- at most three more 4-byte loads, predicated, next to the centre's own.
__device__ __forceinline__ unsigned gm_tile_mask(const double *pay, int tile, int lx, int ly) {
  return reinterpret_cast<const unsigned *>(pay + 4 * (((size_t)tile << (2 * kTileShift)) + ((size_t)ly << kTileShift) + lx) + 3)[0];
}
__device__ __forceinline__ double gm_fresh_value_w1_nbr_tiled(const MapView &m, const double *unk, const int *tiles,
                                                              const GmParams &gp, int cx, int cy, double ox, double oy) {
  const int ix0 = cx + m.origin_x, iy0 = cy + m.origin_y;
  const bool inner = ix0 >= 1 && iy0 >= 1 && ix0 + 1 < m.width && iy0 + 1 < m.height;  // (m.width / height: the extent)
  if (!inner) {
    GmWindow1 w;
    w.issue_occ(m, unk, tiles, cx, cy);
    w.issue_obst(m, unk, gp);
    return w.finish(gp, ox, oy);
  }
  const int txl = (ix0 - 1) >> kTileShift, txh = (ix0 + 1) >> kTileShift, txc = ix0 >> kTileShift;
  const int tyl = (iy0 - 1) >> kTileShift, tyh = (iy0 + 1) >> kTileShift, tyc = iy0 >> kTileShift;
  const int t00 = tiles[tyl * m.pitch + txl], t01 = tiles[tyl * m.pitch + txh];
  const int t10 = tiles[tyh * m.pitch + txl], t11 = tiles[tyh * m.pitch + txh];
  const int lx = ix0 & kTileMask, ly = iy0 & kTileMask;
  const bool cx_lo = txc == txl, cy_lo = tyc == tyl;  // which of the corner tiles holds the centre
  const int tc = cy_lo ? (cx_lo ? t00 : t01) : (cx_lo ? t10 : t11);
  const double *pay = m.payload;
  unsigned m9 = gm_tile_mask(pay, tc, lx, ly);
  const bool rim_x = txl != txh, rim_y = tyl != tyh;  // (the window reaches into another tile column / row)
  // the other tile column / row: the one the centre is NOT in
  unsigned mx = 0u, my = 0u, md = 0u;
  const int tx_other = cy_lo ? (cx_lo ? t01 : t00) : (cx_lo ? t11 : t10);
  const int ty_other = cy_lo ? (cx_lo ? t10 : t11) : (cx_lo ? t00 : t01);
  const int td_other = cy_lo ? (cx_lo ? t11 : t10) : (cx_lo ? t01 : t00);
  const int ox_lx = cx_lo ? 0 : kTileMask, oy_ly = cy_lo ? 0 : kTileMask;  // the cell across: first or last column / row
  if (rim_x) mx = gm_tile_mask(pay, tx_other, ox_lx, ly);
  if (rim_y) my = gm_tile_mask(pay, ty_other, lx, oy_ly);
  if (rim_x && rim_y) md = gm_tile_mask(pay, td_other, ox_lx, oy_ly);
  // that cell's own column (bits 3..5: dy = -1, 0, +1) is this window's column dx = +1 (centre in the low tile) or -1
  if (rim_x) m9 |= ((mx >> 3) & 7u) << (cx_lo ? 6 : 0);
  // ... its own row (bits 1, 4, 7: dx = -1, 0, +1) this window's row dy = +1 or -1 (bits 2, 5, 8 / 0, 3, 6)
  if (rim_y) {
    const unsigned r3 = ((my >> 1) & 1u) | (((my >> 4) & 1u) << 3) | (((my >> 7) & 1u) << 6);
    m9 |= r3 << (cy_lo ? 2 : 0);
  }
  if (rim_x && rim_y) m9 |= ((md >> 4) & 1u) << ((cx_lo ? 6 : 0) + (cy_lo ? 2 : 0));
  double obx9[9], oby9[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    obx9[i] = 0.0;
    oby9[i] = 0.0;
    if ((m9 & (1u << i)) != 0u) {
      const int ix = ix0 + i / 3 - 1, iy = iy0 + i % 3 - 1;  // dx outer, dy inner like the reference
      const bool lo_x = (ix >> kTileShift) == txl, lo_y = (iy >> kTileShift) == tyl;
      const int tile = lo_y ? (lo_x ? t00 : t01) : (lo_x ? t10 : t11);
      const double *c = pay + 4 * (((size_t)tile << (2 * kTileShift)) + ((size_t)(iy & kTileMask) << kTileShift) + (ix & kTileMask));
      obx9[i] = c[1];
      oby9[i] = c[2];
    }
  }
  double best_d2 = __builtin_inf();
  bool any = false;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const double ddx = obx9[i] - ox, ddy = oby9[i] - oy;
    const double d2 = ddx * ddx + ddy * ddy;
    const bool better = (m9 & (1u << i)) != 0u && d2 < best_d2;  // (a NaN distance never wins)
    best_d2 = better ? d2 : best_d2;
    any |= better;
  }
  return gm_value_of(best_d2, any);
}

__device__ __forceinline__ double gm_fresh_value_w1(const MapView &m, const double *unk, const int *tiles,
                                                    const GmParams &gp, int cx, int cy, double ox, double oy) {
  if (m.nbr_ok) {
    return tiles ? gm_fresh_value_w1_nbr_tiled(m, unk, tiles, gp, cx, cy, ox, oy)
                 : gm_fresh_value_w1_nbr(m, unk, gp, cx, cy, ox, oy);
  }
  GmWindow1 w;
  w.issue_occ(m, unk, tiles, cx, cy);
  w.issue_obst(m, unk, gp);
  return w.finish(gp, ox, oy);
}

__device__ __forceinline__ double gm_fresh_value(const MapView &m, const double *unk, const int *tiles,
                                                 const GmParams &gp, int cx, int cy, double ox, double oy) {
  // The value is the maximum over the window's full cells of 1 - (1 - exp(-d^2 / 0.05)), d = distance
  // from the cell's obstacle mean to the beam's end point.  That function falls with d^2, so the
  // maximum belongs to the smallest d^2: the window only tracks that, and ONE exp is evaluated per beam.
  // (An exp per full cell -- up to nine per beam next to a wall, executed by the whole wave as soon as
  // one lane needs it -- was 4.5 of the 10 us of this phase in a lone launch.)
  if (gp.window == 1) return gm_fresh_value_w1(m, unk, tiles, gp, cx, cy, ox, oy);
  double best_d2 = __builtin_inf();
  bool any = false;
  const double4 *cells = reinterpret_cast<const double4 *>(m.payload);
  for (int dx = -gp.window; dx <= gp.window; ++dx) {
    for (int dy = -gp.window; dy <= gp.window; ++dy) {
      const int ix = cx + dx + m.origin_x, iy = cy + dy + m.origin_y;
      const bool inb = (unsigned)ix < (unsigned)m.width && (unsigned)iy < (unsigned)m.height;
      double occ = unk[0], obx = unk[1], oby = unk[2];
      if (inb) {
        size_t at;
        if (tiles) {
          const int tile = tiles[(iy >> kTileShift) * m.pitch + (ix >> kTileShift)];  // pitch = tiles per row
          at = ((size_t)tile << (2 * kTileShift)) + ((size_t)(iy & kTileMask) << kTileShift) + (ix & kTileMask);
        } else {
          at = (size_t)iy * m.pitch + ix;
        }
        const double4 v = cells[at];
        occ = v.x; obx = v.y; oby = v.z;
      }
      if (occ < gp.fullness_th) continue;
      const double ddx = obx - ox, ddy = oby - oy;
      const double d2 = ddx * ddx + ddy * ddy;
      if (d2 < best_d2) {  // (a NaN obstacle never wins, like `best < v` before)
        best_d2 = d2;
        any = true;
      }
    }
  }
  if (!any) return 0.0;
  const double similarity = gm_exp(-best_d2 / 0.05);
  const double v = 1.0 - (1.0 - similarity);
  return 0.0 < v ? v : 0.0;
}

// The mask form in stages of independent loads (see GmWindow1), for callers that interleave two windows: the mask ->
// the obstacle means of the full cells -> the value.  A centre cell on the window's rim has no complete mask: its
// value is made at once, the nine-cell way.
struct GmWindowN {
  unsigned m9;
  bool done;
  double v;
  const double *c0;
  size_t row;
  double obx9[9], oby9[9];
  __device__ __forceinline__ void issue_mask(const MapView &m, const double *unk, const GmParams &gp, int cx, int cy, double ox,
                                             double oy) {
    const int ix0 = cx + m.origin_x, iy0 = cy + m.origin_y;
    done = !(ix0 >= 1 && iy0 >= 1 && ix0 + 1 < m.width && iy0 + 1 < m.height);
    m9 = 0u;
    v = 0.0;
    row = 4 * (size_t)m.pitch;
    c0 = m.payload;
    if (done) {
      GmWindow1 w;
      w.issue_occ(m, unk, nullptr, cx, cy);
      w.issue_obst(m, unk, gp);
      v = w.finish(gp, ox, oy);
    } else {
      c0 = m.payload + row * (size_t)(iy0 - 1) + 4 * (size_t)(ix0 - 1);
      m9 = reinterpret_cast<const unsigned *>(c0 + row + 4 + 3)[0];
    }
  }
  __device__ __forceinline__ void issue_obst() {
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      obx9[i] = 0.0;
      oby9[i] = 0.0;
      if ((m9 & (1u << i)) != 0u) {
        const double *c = c0 + (i % 3) * row + 4 * (i / 3);
        obx9[i] = c[1];
        oby9[i] = c[2];
      }
    }
  }
  __device__ __forceinline__ double finish(double ox, double oy) const {
    if (done) return v;
    double best_d2 = __builtin_inf();
    bool any = false;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const double ddx = obx9[i] - ox, ddy = oby9[i] - oy;
      const double d2 = ddx * ddx + ddy * ddy;
      const bool better = (m9 & (1u << i)) != 0u && d2 < best_d2;
      best_d2 = better ? d2 : best_d2;
      any |= better;
    }
    return gm_value_of(best_d2, any);
  }
};

// ... and ONE cell of such a window, for a helper lane (GmWindowCell)
struct GmWindowCellN {
  bool full, done;
  double dv;
  const double *c;
  __device__ __forceinline__ void issue_mask(const MapView &m, const double *unk, const GmParams &gp, int cx, int cy, int i,
                                             double ox, double oy) {
    const int ix0 = cx + m.origin_x, iy0 = cy + m.origin_y;
    done = !(ix0 >= 1 && iy0 >= 1 && ix0 + 1 < m.width && iy0 + 1 < m.height);
    full = false;
    dv = __builtin_inf();
    c = m.payload;
    if (done) {
      GmWindowCell w;
      w.issue_occ(m, unk, nullptr, cx, cy, i);
      w.issue_obst(m, unk, gp);
      dv = w.d2(gp, ox, oy);
    } else {
      const size_t row = 4 * (size_t)m.pitch;
      const unsigned m9 = reinterpret_cast<const unsigned *>(m.payload + row * (size_t)iy0 + 4 * (size_t)ix0 + 3)[0];
      full = ((m9 >> i) & 1u) != 0u;
      c = m.payload + row * (size_t)(iy0 + i % 3 - 1) + 4 * (size_t)(ix0 + i / 3 - 1);
    }
  }
  double obx, oby;
  __device__ __forceinline__ void issue_obst() {
    obx = 0.0;
    oby = 0.0;
    if (full) {
      obx = c[1];
      oby = c[2];
    }
  }
  __device__ __forceinline__ double d2(double ox, double oy) const {
    if (done) return dv;
    const double ddx = obx - ox, ddy = oby - oy;
    const double v = ddx * ddx + ddy * ddy;
    return (full && v < __builtin_inf()) ? v : __builtin_inf();
  }
};

// One pose scored by ONE workgroup of NT threads (NT >= 256): phase A (end point, window, exp) over all threads --
// beam b goes to thread b % NT --, run resolution (Q19: every maximal run of equal end cells takes the value of its
// first beam) by the beams' own threads, and the canonical sum by the first 256 threads (thread t adds the terms of
// beams t, t + 256, ... in that order, then the wave's tree, then the four waves' partial sums: the one order every
// GMapping path uses).  s_dyn: val[256 KB] | grp_cell int2 [4 KB] | grp_start int [4 KB] | 256 KB doubles: the terms
// (in front of them, during phase A, the cells of the helpers' beams) | NT doubles (NT >= 512); *s_run0 must hold n
// (set before a barrier).  Thread 0 leaves the score in *score_out (its own copy) and the side outputs of the
// cross-pose cache in *gi_out.  r0 / ca0 / sa0: the constants of beam `threadIdx.x`, loaded by the caller ahead of the
// pose.
// (r05: the run resolution used to be the first 256 threads' too, KB beams one after the other with an LDS round
// trip or three in each: 2.7 of a 1024-thread workgroup's 14 us per pose.)
template <int KB, int NT>
__device__ __forceinline__ void gm_score_pose_wide(const MapView &map, const ScanView &scan, const GmParams &gm,
                                                   const int *tiles, const double *s_unknown, double x, double y, double sn,
                                                   double cs, double r0, double ca0, double sa0, double *s_dyn, int *s_run0,
                                                   double *s_part1, GmPoseInfo *gi_out, double *score_out,
                                                   long long *stamp_a = nullptr, int tid = -1) {
  // (tid: the caller's own copy of threadIdx.x -- a persistent kernel hands in one the compiler cannot see through,
  // so that nothing derived from it is hoisted out of its loop and held in registers, hc_resident_gm.hip)
  const int t = tid < 0 ? (int)threadIdx.x : tid;
  const int lane = t & 63, wave = t >> 6;
  const int n = scan.n;
  const int G = (n + 63) >> 6;
  constexpr int R = (KB * kGmBlock + NT - 1) / NT;  // beams per thread: b = t + NT r
  double *s_val = s_dyn;
  int2 *s_grp_cell = reinterpret_cast<int2 *>(s_dyn + (size_t)KB * kGmBlock);
  int *s_grp_start = reinterpret_cast<int *>(s_grp_cell + 4 * KB);
  double *s_term = reinterpret_cast<double *>(s_grp_start + 4 * KB);
  int2 *s_hcell = reinterpret_cast<int2 *>(s_term);  // (phase A only: the helpers' beams' cells)
  double *s_sur = s_term + (size_t)KB * kGmBlock;    // NT doubles (kGmHelperDoubles<NT>)
  int &s_run0_len = *s_run0;
  const double scale = map.scale, inv_scale = map.inv_scale;
  int cxr[R], cyr[R];
#pragma unroll
  for (int r = 0; r < R; ++r) cxr[r] = cyr[r] = 0;
  // phase A over all threads.  A scan a little longer than the workgroup -- 1080 beams on 1024 threads -- would send
  // the first wave through the phase TWICE for its 56 surplus beams (another end point, another two round trips of
  // gathers).  Instead nine HELPER lanes per surplus beam fetch one cell of its window each, next to their own beam's
  // window and in the same round trips; the beam's value is made from their nine distances behind the barrier (the
  // same selection: the smallest distance to a full cell).
  const int surplus = n - NT;
  // (through a pool's tiles with masks the surplus beams take a second round of the first wave instead)
  const bool helpers = NT >= 512 && gm.window == 1 && surplus > 0 && 9 * surplus <= NT && !(map.nbr_ok && tiles);
  if (helpers) {
    const double c = cs * ca0 - sn * sa0;
    const double s = sn * ca0 + cs * sa0;
    const double wx = x + r0 * c;
    const double wy = y + r0 * s;
    const int cx = to_cell(wx, scale, inv_scale), cy = to_cell(wy, scale, inv_scale);
    const bool hlp = t < 9 * surplus;
    const int hb = NT + t / 9, hi = t - 9 * (t / 9);
    double hwx = 0.0, hwy = 0.0;
    int hcx = 0, hcy = 0;
    if (hlp) {
      const double hr = scan.range[hb], hca = scan.cos_a[hb], hsa = scan.sin_a[hb];
      const double hc_ = cs * hca - sn * hsa;
      const double hs_ = sn * hca + cs * hsa;
      hwx = x + hr * hc_;
      hwy = y + hr * hs_;
      hcx = to_cell(hwx, scale, inv_scale);
      hcy = to_cell(hwy, scale, inv_scale);
    }
    double own_v, sur_d2 = __builtin_inf();
    if (map.nbr_ok && !tiles) {
      GmWindowN own;
      GmWindowCellN cell;
      own.issue_mask(map, s_unknown, gm, cx, cy, wx, wy);
      if (hlp) cell.issue_mask(map, s_unknown, gm, hcx, hcy, hi, hwx, hwy);
      own.issue_obst();
      if (hlp) cell.issue_obst();
      own_v = own.finish(wx, wy);
      if (hlp) sur_d2 = cell.d2(hwx, hwy);
    } else {
      GmWindow1 own;
      GmWindowCell cell;
      own.issue_occ(map, s_unknown, tiles, cx, cy);
      if (hlp) cell.issue_occ(map, s_unknown, tiles, hcx, hcy, hi);
      own.issue_obst(map, s_unknown, gm);
      if (hlp) cell.issue_obst(map, s_unknown, gm);
      own_v = own.finish(gm, wx, wy);
      if (hlp) sur_d2 = cell.d2(gm, hwx, hwy);
    }
    s_val[t] = own_v;
    cxr[0] = cx;
    cyr[0] = cy;
    if (lane == 63) s_grp_cell[wave] = make_int2(cx, cy);  // (NT < n: beam t is never the last one)
    if (hlp) {
      s_sur[t] = sur_d2;
      if (hi == 0) s_hcell[hb - NT] = make_int2(hcx, hcy);
    }
  } else {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int b = t + NT * r;
      if (b < n) {
        const double rr = r == 0 ? r0 : scan.range[b], ca = r == 0 ? ca0 : scan.cos_a[b], sa = r == 0 ? sa0 : scan.sin_a[b];
        const double c = cs * ca - sn * sa;
        const double s = sn * ca + cs * sa;
        const double wx = x + rr * c;
        const double wy = y + rr * s;
        const int cx = to_cell(wx, scale, inv_scale), cy = to_cell(wy, scale, inv_scale);
        s_val[b] = gm_fresh_value(map, s_unknown, tiles, gm, cx, cy, wx, wy);
        cxr[r] = cx;
        cyr[r] = cy;
        if (lane == 63 || b == n - 1) s_grp_cell[b >> 6] = make_int2(cx, cy);
      }
    }
  }
  __syncthreads();
  if (stamp_a) *stamp_a = wall_clock64();  // (tools/hc_chain_stamps.py: the end of phase A)
  // ---- runs: which beams start one.  Beam b's group = b >> 6 = its wave in its round; the cell in front of a
  // group's first beam is the last one of the group before (s_grp_cell).  The weights and factors of this thread's
  // beams are asked for here, two barriers ahead of their use.
  double bw[R], bf[R];
  unsigned long long mask[R];
  if (R >= 2 && helpers && t < surplus) {
    const int2 hc = s_hcell[t];
    cxr[R >= 2 ? 1 : 0] = hc.x;
    cyr[R >= 2 ? 1 : 0] = hc.y;
    double best_d2 = __builtin_inf();
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const double d2 = s_sur[9 * t + i];
      best_d2 = d2 < best_d2 ? d2 : best_d2;
    }
    s_val[NT + t] = gm_value_of(best_d2, best_d2 < __builtin_inf());  // (read behind the next barrier)
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int b = t + NT * r;
    bw[r] = bf[r] = 0.0;
    if (b < n) {
      bw[r] = scan.weight[b];
      bf[r] = scan.factor[b];
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int b = t + NT * r;
    const int g = b >> 6;
    int pcx = __shfl_up(cxr[r], 1, 64), pcy = __shfl_up(cyr[r], 1, 64);
    int2 pc = s_grp_cell[(g > 0 && g < G) ? g - 1 : 0];  // (one address per wave)
    // (the helpers' branch of phase A writes s_grp_cell for the workgroup's OWN beams only -- groups
    // 0 .. NT / 64 - 1.  With more than 64 surplus beams, 1089 .. 1137 beams on 1024 threads, the first beam of a later
    // surplus group found no cell in front of it there.  The cell of the surplus beam before it is in s_hcell, written in
    // phase A in front of the first barrier and not overwritten before the next one.)
    if (helpers && lane == 0 && b > NT && b < n) pc = s_hcell[b - NT - 1];
    if (lane == 0 && g > 0) {
      pcx = pc.x;
      pcy = pc.y;
    }
    const bool start = (b < n) && (b == 0 || pcx != cxr[r] || pcy != cyr[r]);
    mask[r] = __ballot(start);
    if (lane == 0 && g < G) s_grp_start[g] = mask[r] ? (64 * g + 63 - __clzll(mask[r])) : -1;
    const unsigned long long later = g == 0 ? mask[r] & ~1ull : mask[r];  // see k_score_gmapping
    if (lane == 0 && later) atomicMin(&s_run0_len, 64 * g + __ffsll((long long)later) - 1);
  }
  __syncthreads();
  // ---- every beam's term: the value of its run's first beam x weight x factor
  double acc = 0.0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int b = t + NT * r;
    if (b < n) {
      const int g = b >> 6;
      const unsigned long long upto = mask[r] & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
      int head;
      if (upto) {
        head = 64 * g + 63 - __clzll(upto);
      } else {
        int gg = g - 1;
        head = s_grp_start[gg];
        while (head < 0) head = s_grp_start[--gg];  // beam 0 is always a start
      }
      const double v = s_val[head];
      const double term = v * bw[r] * bf[r];
      if (NT == kGmBlock) acc = acc + term;  // (this thread's beams ARE its canonical ones, in order)
      else s_term[b] = term;
      if (b == n - 1 && gi_out) {
        GmPoseInfo &gi = *gi_out;
        gi.last_cx = cxr[r];
        gi.last_cy = cyr[r];
        gi.last_v = v;
        gi.last_head = head;
      }
      if (b == 0 && gi_out) {
        GmPoseInfo &gi = *gi_out;
        gi.first_cx = cxr[r];
        gi.first_cy = cyr[r];
        gi.v0 = v;
      }
    }
  }
  const bool act = t < kGmBlock;
  if (NT != kGmBlock) {
    __syncthreads();
    if (act) {
#pragma unroll
      for (int k = 0; k < KB; ++k) {
        const int b = t + kGmBlock * k;
        if (b < n) acc = acc + s_term[b];
      }
    }
  }
  acc = wave_xor_sum(acc);
  if (act && lane == 0) s_part1[wave] = acc;
  __syncthreads();
  if (t == 0) {
    if (gi_out) gi_out->run0_len = s_run0_len;
    const double total = (s_part1[0] + s_part1[1]) + (s_part1[2] + s_part1[3]);
    *score_out = (scan.tot_w == 0.0) ? __builtin_nan("") : total / scan.tot_w;
  }
}

}  // namespace slamhip
