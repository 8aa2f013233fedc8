// exact_kernels.hip -- the libm-exact modes: scoring whose transcendental functions return the bits
// of the reference host's glibc (csrc/libm_exact.h), for the two configurations whose per-beam value depends on them:
//
//   * SLAMHIP_POSE_TRIG_RAW_EXACT -- the reference's DEFAULT trig provider: RawTrigonometryProvider evaluates
//     std::cos / std::sin(theta + a) per beam and pose (src/core/trigonometry_utils.h:17-35, selected by
//     use_trig_cache = false, src/ros/init_utils.h:56-58); every other mode evaluates the cached provider's angle
//     addition, whose end points differ in the last place.  k_exact_beam_trig tabulates cos / sin(theta_p + a_b) for a
//     batch of poses; the scoring kernels then run pose by pose over that pose's table with the pose rotation set
//     to the identity (cs = 1, sn = 0: c = 1 ca - 0 sa = ca and s = 0 ca + 1 sa = sa exactly), i.e. unchanged.
//   * the GMapping OOPE in strict mode (SLAMHIP_SUM_SEQUENTIAL): k_score_gmapping_exact restates
//     GmappingOccupancyObservationPE::probability (src/slams/gmapping/gmapping_occupancy_observation_pe.h:17-38) and
//     GmappingBaseCell::discrepancy (gmapping_grid_cell.h:35-38: 1 - std::exp(-d^2 / 0.05)) the plain way -- an exp per
//     full cell in the reference's window order, the cell cache applied beam after beam and carried from pose to pose
//     in call order, the sum in beam order (weighted_mean_point_probability_spe.h:101-132) -- with glibc's exp.
//
// Neither is a fast path: they are what the default mode is checked against, and what a caller asks for who needs
// the reference's bits.  FMA: which build of glibc's functions the host runs (slamhip_libm_variant).
#include "gm_score_device.h"
#include "libm_exact.h"

namespace slamhip {

template <bool FMA>
__global__ __launch_bounds__(256) void k_exact_beam_trig(const double *__restrict__ poses, int n_poses,
                                                        const double *__restrict__ angle, int n, size_t stride,
                                                        double *__restrict__ out_cos, double *__restrict__ out_sin,
                                                        double *__restrict__ identity_sc) {
  const int p = blockIdx.y;
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (p >= n_poses) return;
  if (b == 0) {
    identity_sc[2 * p] = 0.0;      // sin
    identity_sc[2 * p + 1] = 1.0;  // cos
  }
  if (b < n) {
    const double x = poses[3 * p + 2] + angle[b];  // _base_angle + angle_rad (trigonometry_utils.h:22,26)
    out_cos[(size_t)p * stride + b] = libm_exact::cos_<FMA>(x);
    out_sin[(size_t)p * stride + b] = libm_exact::sin_<FMA>(x);
  }
}

hipError_t launch_exact_beam_trig(bool fma, const double *poses, int n_poses, const double *d_angle, int n, size_t stride,
                                  double *d_cos, double *d_sin, double *d_identity_sc, hipStream_t stream) {
  const dim3 grid((n + 255) / 256, n_poses);
  if (fma)
    hipLaunchKernelGGL(k_exact_beam_trig<true>, grid, dim3(256), 0, stream, poses, n_poses, d_angle, n, stride, d_cos, d_sin,
                       d_identity_sc);
  else
    hipLaunchKernelGGL(k_exact_beam_trig<false>, grid, dim3(256), 0, stream, poses, n_poses, d_angle, n, stride, d_cos, d_sin,
                       d_identity_sc);
  return hipGetLastError();
}

// one cell of the window: (prob_occ, obst.x, obst.y) or the prototype outside the map
__device__ __forceinline__ double4 gm_exact_cell(const MapView &m, const int *tiles, int ix, int iy) {
  double4 v = make_double4(m.unknown[0], m.unknown[1], m.unknown[2], 0.0);
  if ((unsigned)ix < (unsigned)m.width && (unsigned)iy < (unsigned)m.height) {
    size_t at;
    if (tiles) {
      const int tile = tiles[(iy >> kTileShift) * m.pitch + (ix >> kTileShift)];  // pitch = tiles per row
      at = ((size_t)tile << (2 * kTileShift)) + ((size_t)(iy & kTileMask) << kTileShift) + (ix & kTileMask);
    } else {
      at = (size_t)iy * m.pitch + ix;
    }
    const double *c = m.payload + 4 * at;
    v.x = c[0];
    v.y = c[1];
    v.z = c[2];
  }
  return v;
}

// ONE workgroup walks the poses of the call in order.  Per pose: phase A, all threads -- end point (trig_mode 0: the
// pose's sin / cos from pose_sc or the device's sincos and the angle addition; 1: glibc's cos / sin(theta + a)), its cell,
// the value a cache miss would compute; phase B, thread 0 -- the reference's loop over the beams: cache hit or miss,
// total += p w factor.  `cache` (cell, prob; prob -1 = empty) lives in device memory: the launches of a stream see the
// reference's ONE cache object in call order.
struct GmExactCache {
  int cx, cy;
  double prob;
};

template <bool FMA>
__global__ __launch_bounds__(256) void k_score_gmapping_exact(ScoreArgs a, const double *__restrict__ angle, int raw_trig,
                                                             GmExactCache *cache) {
  extern __shared__ double s_dyn[];  // val[n] | cell int2 [n]
  __shared__ double s_pose[4];
  const int n = a.scan.n;
  double *s_val = s_dyn;
  int2 *s_cell = reinterpret_cast<int2 *>(s_dyn + n);
  const int t = threadIdx.x;
  const double scale = a.map.scale;
  for (int p = 0; p < a.n_poses; ++p) {
    if (t == 0) {
      const double th = a.poses[3 * p + 2];
      double sn = 0.0, cs = 1.0;
      if (!raw_trig) {
        if (a.pose_sc) {
          sn = a.pose_sc[2 * p];
          cs = a.pose_sc[2 * p + 1];
        } else {
          sincos(th, &sn, &cs);
        }
      }
      s_pose[0] = a.poses[3 * p];
      s_pose[1] = a.poses[3 * p + 1];
      s_pose[2] = sn;
      s_pose[3] = cs;
    }
    __syncthreads();
    const double x = s_pose[0], y = s_pose[1], sn = s_pose[2], cs = s_pose[3];
    const double th = a.poses[3 * p + 2];
    const int *tiles = a.tables ? a.tables + (size_t)a.pose_slot[p] * a.table_stride : nullptr;
    for (int b = t; b < n; b += 256) {
      double c, s;
      if (raw_trig) {
        const double ang = th + angle[b];
        c = libm_exact::cos_<FMA>(ang);
        s = libm_exact::sin_<FMA>(ang);
      } else {
        const double ca = a.scan.cos_a[b], sa = a.scan.sin_a[b];
        c = cs * ca - sn * sa;
        s = sn * ca + cs * sa;
      }
      const double r = a.scan.range[b];
      const double wx = x + r * c;
      const double wy = y + r * s;
      const int cx = (int)floor(wx / scale), cy = (int)floor(wy / scale);  // regular_squares_grid.h:40-46
      double best = 0.0;
      for (int dx = -a.gm.window; dx <= a.gm.window; ++dx)
        for (int dy = -a.gm.window; dy <= a.gm.window; ++dy) {
          const double4 cell = gm_exact_cell(a.map, tiles, cx + dx + a.map.origin_x, cy + dy + a.map.origin_y);
          if (cell.x < a.gm.fullness_th) continue;
          const double ddx = cell.y - wx, ddy = cell.z - wy;
          const double d2 = ddx * ddx + ddy * ddy;
          const double similarity = libm_exact::exp_<FMA>(-d2 / 0.05);
          const double v = 1.0 - (1.0 - similarity);
          best = best < v ? v : best;  // std::max(best_prob, v)
        }
      s_val[b] = best;
      s_cell[b] = make_int2(cx, cy);
    }
    __syncthreads();
    if (t == 0) {
      int ccx = cache->cx, ccy = cache->cy;
      double cprob = cache->prob;
      double total = 0.0;
      for (int b = 0; b < n; ++b) {
        const int2 cell = s_cell[b];
        double pr;
        if (cell.x == ccx && cell.y == ccy && cprob != -1.0) {
          pr = cprob;
        } else {
          pr = s_val[b];
          ccx = cell.x;
          ccy = cell.y;
          cprob = pr;
        }
        total = total + pr * a.scan.weight[b] * a.scan.factor[b];
      }
      cache->cx = ccx;
      cache->cy = ccy;
      cache->prob = cprob;
      a.scores[p] = (a.scan.tot_w == 0.0) ? __builtin_nan("") : total / a.scan.tot_w;
    }
    __syncthreads();
  }
}

hipError_t launch_score_gmapping_exact(bool fma, const ScoreArgs &a, const double *d_angle, int raw_trig, void *d_cache,
                                       hipStream_t stream) {
  const size_t shm = (size_t)a.scan.n * (sizeof(double) + sizeof(int2));
  if (shm > 60 * 1024) return hipErrorInvalidValue;
  if (fma)
    hipLaunchKernelGGL(k_score_gmapping_exact<true>, dim3(1), dim3(256), shm, stream, a, d_angle, raw_trig,
                       static_cast<GmExactCache *>(d_cache));
  else
    hipLaunchKernelGGL(k_score_gmapping_exact<false>, dim3(1), dim3(256), shm, stream, a, d_angle, raw_trig,
                       static_cast<GmExactCache *>(d_cache));
  return hipGetLastError();
}

// ---- the device's restated functions, for the tests (tests/test_gpu_libm_exact.py: device == host, bit for bit) ------
template <bool FMA>
__global__ __launch_bounds__(256) void k_libm_eval(int fn, const double *__restrict__ x, double *__restrict__ out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  out[i] = fn == 0 ? libm_exact::sin_<FMA>(x[i]) : (fn == 1 ? libm_exact::cos_<FMA>(x[i]) : libm_exact::exp_<FMA>(x[i]));
}
hipError_t launch_libm_eval(bool fma, int fn, const double *d_x, double *d_out, int n, hipStream_t stream) {
  if (fma) hipLaunchKernelGGL(k_libm_eval<true>, dim3((n + 255) / 256), dim3(256), 0, stream, fn, d_x, d_out, n);
  else hipLaunchKernelGGL(k_libm_eval<false>, dim3((n + 255) / 256), dim3(256), 0, stream, fn, d_x, d_out, n);
  return hipGetLastError();
}

}  // namespace slamhip
