// score_device.h -- device functions shared by the scoring kernels (score_kernels.hip) and the
// device-resident hill-climbing chain (hc_chain.hip): one definition, so a pose scored by either
// kernel gets the same bits.
//   RegularSquaresGrid::world_to_cell      src/core/maps/regular_squares_grid.h:40-46
//   ObstacleBasedOccupancyObservationPE    src/core/scan_matchers/occupancy_observation_probability.h:12-27
//   DiscrepancyOIE / OccupancyOIE          src/core/scan_matchers/observation_impact_estimators.h:14-28
//   GridCell::discrepancy                  src/core/maps/grid_cell.h:33-35
//   TbmBaseCell::discrepancy + conjunctive src/core/maps/tbm_grid_cells.h:21-35,
//                                          src/core/maps/transferable_belief_model.h:102-143
#pragma once

#include "slamhip_internal.h"

namespace slamhip {

// ---- helpers ---------------------------------------------------------------------------------
__device__ __forceinline__ double wave_xor_sum(double v) {
  // fixed butterfly: every lane ends with the same bits (a+b == b+a in IEEE)
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = v + __shfl_xor(v, off, 64);
  return v;
}

// ---- term-vector fingerprints (the matchers' checked default mode) -------------------------------
// h = sum over beams of lo32(term) * k_lo(b) + hi32(term) * k_hi(b) mod 2^64 with odd per-beam multipliers
// k(b) = (2b + 1) * C mod 2^32: integer arithmetic, so every order of adding it up gives the same value; one
// differing term changes it for certain, terms exchanged between beams change it unless the multipliers conspire.
__device__ __forceinline__ unsigned long long term_fingerprint(double term, unsigned k_lo, unsigned k_hi) {
  const unsigned long long bits = (unsigned long long)__double_as_longlong(term);
  return (unsigned long long)(unsigned)bits * k_lo + (unsigned long long)(unsigned)(bits >> 32) * k_hi;
}
__device__ __forceinline__ unsigned fold_fingerprint(unsigned long long h) { return (unsigned)(h >> 32) ^ (unsigned)h; }
// wave_xor_sum's fixed butterfly with the fingerprint's exchanges riding along (one LDS-crossbar latency per step
// for both)
__device__ __forceinline__ void wave_xor_sum_with(double &v, unsigned long long &h) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const double o = __shfl_xor(v, off, 64);
    const unsigned ol = (unsigned)__shfl_xor((int)(unsigned)h, off, 64);
    const unsigned oh = (unsigned)__shfl_xor((int)(unsigned)(h >> 32), off, 64);
    v = v + o;
    h += ((unsigned long long)oh << 32) | ol;
  }
}

// world_to_cell: int(floor(x / scale)) with a TRUE division (Q15: multiplying by 1/scale flips
// cells at boundaries).  The division is the most expensive thing in the per-beam body, so it is
// only executed when it can matter: with t = v/scale, RN(t) is within 2^-53 |t| of t and
// q = RN(v * RN(1/scale)) within 2^-52 |t|, hence no integer separates q from RN(t) unless q lies
// within 1.5 * 2^-52 |q| of one.  Outside a 2^-49 |q| band around integers floor(q) IS the
// reference's cell; inside it (endpoints exactly on cell boundaries do occur, e.g. the reference's
// HC smoke fixture) the true quotient is evaluated.  The absolute term sends underflowing products
// to the exact path as well.
__device__ __forceinline__ int to_cell(double v, double scale, double inv_scale) {
  const double q = v * inv_scale;
  const double f = floor(q);
  const double d = q - f;
  const double tol = fabs(q) * 0x1p-49 + 0x1p-1000;
  if (__builtin_expect(d < tol || (1.0 - d) < tol, 0)) return (int)floor(v / scale);
  return (int)f;
}

// payload of cell (cx, cy) -- or of the prototype cell outside the window (UnboundedPlainGridMap::operator[],
// plain_grid_map.h:69-73); OCC uses .x only.  Split from the probability so that a thread can have the
// gathers of several beams in flight before it needs the first value.
template <int MODEL>
__device__ __forceinline__ double4 load_cell(const MapView &m, int cx, int cy) {
  const int ix = cx + m.origin_x, iy = cy + m.origin_y;
  const bool inb = (unsigned)ix < (unsigned)m.width && (unsigned)iy < (unsigned)m.height;
  double4 v = make_double4(m.unknown[0], m.unknown[1], m.unknown[2], m.unknown[3]);
  if (MODEL == SLAMHIP_CELL_OCC) {
    if (inb) v.x = m.payload[(size_t)iy * m.pitch + ix];
  } else {
    if (inb) v = *(reinterpret_cast<const double4 *>(m.payload) + ((size_t)iy * m.pitch + ix));
  }
  return v;
}

template <int MODEL>
__device__ __forceinline__ double cell_probability(int oie, const double4 &v) {
  if (MODEL == SLAMHIP_CELL_OCC) {
    const double occ = v.x;
    if (oie == SLAMHIP_OIE_OCCUPANCY) return occ;
    return 1.0 - fabs(occ - 1.0);
  } else {
    const double U = v.x, E = v.y, O = v.z, Cc = v.w;
    // that = aoo2tbm(obstacle AOO) = (u,e,o,c) = (0,0,1,0); conjunctive(that, cell) before
    // normalisation = (0, 0, U+O, E+C); normalize() divides by the total mass.
    const double d_occ = fabs(1.0 - O);
    const double t2 = U + O, t3 = E + Cc;
    const double tot = t2 + t3;
    const double conflict = (tot == 0.0) ? 0.0 : t3 / tot;
    const double unknown = U / 2.0;
    const double known = 1 - unknown;
    const double known_discrepancy = known * (conflict + d_occ) / 2.0;
    return 1.0 - (unknown / 2 + known_discrepancy);
  }
}

template <int MODEL>
__device__ __forceinline__ double point_probability(const MapView &m, int oie, int cx, int cy) {
  return cell_probability<MODEL>(oie, load_cell<MODEL>(m, cx, cy));
}

// one beam of WeightedMeanPointProbabilitySPE::estimate_scan_probability
// (weighted_mean_point_probability_spe.h:108-124): scan point moved to the pose with the cached-provider
// angle addition (sensor_data.h:83-88, trigonometry_utils.h:45-55), its cell, the cell's probability,
// times weight times factor
template <int MODEL>
__device__ __forceinline__ double4 beam_cell(const MapView &m, double x, double y, double sn, double cs, double r,
                                             double ca, double sa) {
  const double c = cs * ca - sn * sa;
  const double s = sn * ca + cs * sa;
  const double wx = x + r * c;
  const double wy = y + r * s;
  return load_cell<MODEL>(m, to_cell(wx, m.scale, m.inv_scale), to_cell(wy, m.scale, m.inv_scale));
}
template <int MODEL>
__device__ __forceinline__ double beam_term(const MapView &m, int oie, double x, double y, double sn, double cs,
                                            double r, double ca, double sa, double w, double f) {
  const double pr = cell_probability<MODEL>(oie, beam_cell<MODEL>(m, x, y, sn, cs, r, ca, sa));
  return pr * w * f;
}

}  // namespace slamhip
