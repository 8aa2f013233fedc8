// score_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X, CDNA4, wave64).
//
// K1  k_score_point     poses x beams, 1-cell ("obstacle") OOPE over OCC / TBM payloads
// K3  k_score_gmapping  poses x beams, 3x3 Gaussian-endpoint OOPE + run-cache resolution
// K1s k_sum_sequential  optional second pass: the reference's beam-order sum, bit-exact
// plus map-mirror maintenance kernels (fill / repack / scatter).
//
// What they restate (paths relative to the reference root):
//   WeightedMeanPointProbabilitySPE::estimate_scan_probability
//       src/core/scan_matchers/weighted_mean_point_probability_spe.h:97-133
//   ScanPoint2D::move_origin + CachedTrigonometryProvider angle addition
//       src/core/states/sensor_data.h:83-88, src/core/trigonometry_utils.h:45-55
//   RegularSquaresGrid::world_to_cell      src/core/maps/regular_squares_grid.h:40-46
//   ObstacleBasedOccupancyObservationPE    src/core/scan_matchers/occupancy_observation_probability.h:12-27
//   DiscrepancyOIE / OccupancyOIE          src/core/scan_matchers/observation_impact_estimators.h:14-28
//   GridCell::discrepancy                  src/core/maps/grid_cell.h:33-35
//   TbmBaseCell::discrepancy + conjunctive src/core/maps/tbm_grid_cells.h:21-35,
//                                          src/core/maps/transferable_belief_model.h:102-143
//   GmappingOccupancyObservationPE         src/slams/gmapping/gmapping_occupancy_observation_pe.h:17-38
//   GmappingBaseCell::discrepancy          src/slams/gmapping/gmapping_grid_cell.h:35-38
//
// Design (DESIGN.md has the long form):
//  * gather + reduction, no MFMA.  All arithmetic is FP64 in the reference's operation order;
//    this file is compiled with -ffp-contract=off so no FMA is formed (the reference's x86-64
//    build has none), which makes per-beam terms bit-identical to the CPU path.
//  * one workgroup = 256 threads = 4 wave64; thread t owns beams t, t+256, ... whose constants
//    (range, cos a, sin a, weight, factor) are loaded ONCE per workgroup into VGPRs with coalesced
//    512-byte-per-wave loads and reused for every pose the workgroup scores; pose constants
//    (x, y, sin th, cos th) are staged in LDS (sincos evaluated once per pose, not per beam).
//  * per-pose sum in a CANONICAL order that does not depend on the launch shape: 256 strided
//    partials (thread t adds its beams in ascending order), a wave64 xor-butterfly (32..1) and
//    (g0+g1)+(g2+g3) across the 4 waves through LDS.  Two launches that score the same pose
//    return bit-identical values, so the matcher's strict `best < candidate` test behaves like
//    the reference's on ties.
//  * map window = pitched row-major HBM array; out-of-window reads return the prototype payload
//    (UnboundedPlainGridMap::operator[], src/core/maps/plain_grid_map.h:69-73).

#include <hip/hip_ext.h>

#include <algorithm>

#include "gm_score_device.h"
#include "score_device.h"

namespace slamhip {

static constexpr int kBlock = 256;
static constexpr int kMaxPosesPerBlock = 16;

// Completion signal for the low-latency host path: scoring kernels write their results straight
// into pinned, coherent host memory with plain stores; this 1-thread kernel, queued right behind
// them on the same stream, publishes the launch sequence number where the host spins.  The kernel
// boundary orders it after every store of the preceding kernel (measured on MI355X: 12 us per
// round trip against 15 us for hipStreamSynchronize and 65 us for per-workgroup system fences;
// tools/latency_probe.hip, 0 stale words in 2000 x 768 hand-offs).
__global__ void k_publish(unsigned *flag, unsigned seq) {
  __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// testing: keeps the stream busy for `ticks` of the 100 MHz wall clock (what a collective whose peer has died looks
// like to a waiting host -- bounded, so that it ends by itself)
__global__ void k_stall(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
hipError_t launch_stall(int ms, hipStream_t stream) {
  hipLaunchKernelGGL(k_stall, dim3(1), dim3(1), 0, stream, (long long)ms * 100000ll);
  return hipGetLastError();
}

hipError_t launch_publish(unsigned *flag, unsigned seq, hipStream_t stream) {
  hipLaunchKernelGGL(k_publish, dim3(1), dim3(1), 0, stream, flag, seq);
  return hipGetLastError();
}

// The scan's way to HBM: a kernel PULLS the packed scan out of pinned host memory (16 bytes per thread over PCIe)
// instead of an SDMA copy queued in front of the match -- hipMemcpyAsync costs the host ~6 us and the stream ~10 us
// before the next kernel starts, this launch ~2.5 and ~4.  The last workgroup through tells the host that the staging
// buffer may be refilled (it has been READ, which is all the host needs to know).
__global__ __launch_bounds__(256) void k_scan_pull(const double2 *__restrict__ src, double2 *__restrict__ dst, int n2,
                                                   unsigned *counter, unsigned *h_flag, unsigned seq) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n2) dst[i] = src[i];
  __syncthreads();  // (every load of the workgroup has returned: its values went into stores)
  if (threadIdx.x == 0) {
    const unsigned before = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (before + 1u == gridDim.x) {
      __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(h_flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

hipError_t launch_scan_pull(const double *h_src, double *d_dst, size_t n_doubles, unsigned *counter, unsigned *h_flag,
                            unsigned seq, hipStream_t stream) {
  const int n2 = (int)((n_doubles + 1) / 2);
  hipLaunchKernelGGL(k_scan_pull, dim3((n2 + 255) / 256), dim3(256), 0, stream,
                     reinterpret_cast<const double2 *>(h_src), reinterpret_cast<double2 *>(d_dst), n2, counter, h_flag,
                     seq);
  return hipGetLastError();
}

// The same for a launch's small argument blocks (job tables, initial poses, a zeroed counter): ONE pull of the pinned
// block in front of the launch instead of several hipMemcpyAsync / hipMemsetAsync calls (each ~6 us of host time and
// ~10 us of stream time, see above).  The host rewrites the block only after the launch that follows has reported its
// results, so nobody needs to be told that it has been read.
__global__ __launch_bounds__(256) void k_block_pull(const uint4 *__restrict__ src, uint4 *__restrict__ dst, int n16) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n16) dst[i] = src[i];
}

hipError_t launch_block_pull(const void *h_src, void *d_dst, size_t bytes, hipStream_t stream) {
  const int n16 = (int)((bytes + 15) / 16);
  if (n16 <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_block_pull, dim3((n16 + 255) / 256), dim3(256), 0, stream, reinterpret_cast<const uint4 *>(h_src),
                     reinterpret_cast<uint4 *>(d_dst), n16);
  return hipGetLastError();
}

// ---- K1 ----------------------------------------------------------------------------------------
// KB > 0: beams per thread known at compile time (n <= 256*KB), constants live in VGPRs.
// KB == 0: generic (any n): constants re-read from L1/L2 in the pose loop.
// FPRINT: a 64-bit fingerprint of every pose's term vector goes out next to its score (the matchers' checked
// default mode: two poses with equal fingerprints add up the same terms, whatever the order of the sum)
template <int MODEL, int KB, bool WRITE_TERMS, bool FPRINT>
__global__ __launch_bounds__(kBlock) void k_score_point(ScoreArgs a) {
  __shared__ double s_pose[kMaxPosesPerBlock][4];
  __shared__ double s_part[kMaxPosesPerBlock][4];
  __shared__ unsigned long long s_hpart[FPRINT ? kMaxPosesPerBlock : 1][4];
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int n = a.scan.n;
  const int p0 = blockIdx.x * a.poses_per_block;
  const int npb = min(a.poses_per_block, a.n_poses - p0);

  // the pose comes over PCIe from the pinned staging buffer: its loads are issued first and the scan
  // constants' (HBM) right behind them, so the two latencies overlap; sincos waits for the pose only
  double pose_x = 0.0, pose_y = 0.0, pose_th = 0.0, pose_sn = 0.0, pose_cs = 0.0;
  if (t < npb) {
    const int p = p0 + t;
    pose_th = a.poses[3 * p + 2];
    pose_x = a.poses[3 * p];
    pose_y = a.poses[3 * p + 1];
    if (a.pose_sc) {
      pose_sn = a.pose_sc[2 * p];
      pose_cs = a.pose_sc[2 * p + 1];
    }
  }

  constexpr int KR = KB > 0 ? KB : 1;
  double br[KR], bc[KR], bs[KR], bw[KR], bf[KR];
  if (KB > 0) {
#pragma unroll
    for (int k = 0; k < KR; ++k) {
      const int b = t + kBlock * k;
      const bool ok = b < n;
      br[k] = ok ? a.scan.range[b] : 0.0;
      bc[k] = ok ? a.scan.cos_a[b] : 0.0;
      bs[k] = ok ? a.scan.sin_a[b] : 0.0;
      bw[k] = ok ? a.scan.weight[b] : 0.0;
      bf[k] = ok ? a.scan.factor[b] : 0.0;
    }
  }
  if (t < npb) {
    if (!a.pose_sc) sincos(pose_th, &pose_sn, &pose_cs);
    s_pose[t][0] = pose_x;
    s_pose[t][1] = pose_y;
    s_pose[t][2] = pose_sn;
    s_pose[t][3] = pose_cs;
  }
  __syncthreads();

  const double scale = a.map.scale, inv_scale = a.map.inv_scale;
  // position-sensitive multilinear form of the terms' 32-bit halves, odd per-beam multipliers (hc_chain.hip's)
  const unsigned fk_lo = (2u * (unsigned)t + 1u) * 0x9E3779B1u, fk_hi = (2u * (unsigned)t + 1u) * 0x85EBCA6Bu;
  for (int j = 0; j < npb; ++j) {
    const double x = s_pose[j][0], y = s_pose[j][1], sn = s_pose[j][2], cs = s_pose[j][3];
    double acc = 0.0;
    unsigned long long h = 0ull;
    if (KB > 0) {
#pragma unroll
      for (int k = 0; k < KR; ++k) {
        const int b = t + kBlock * k;
        if (b < n) {
          const double c = cs * bc[k] - sn * bs[k];
          const double s = sn * bc[k] + cs * bs[k];
          const double wx = x + br[k] * c;
          const double wy = y + br[k] * s;
          const double pr = point_probability<MODEL>(a.map, a.oie, to_cell(wx, scale, inv_scale), to_cell(wy, scale, inv_scale));
          const double term = pr * bw[k] * bf[k];
          if (WRITE_TERMS) a.terms[(size_t)(p0 + j) * n + b] = term;
          acc = acc + term;
          if (FPRINT) h += term_fingerprint(term, fk_lo + (unsigned)k * (2u * kBlock * 0x9E3779B1u),
                                            fk_hi + (unsigned)k * (2u * kBlock * 0x85EBCA6Bu));
        }
      }
    } else {
      for (int b = t; b < n; b += kBlock) {
        const double ca = a.scan.cos_a[b], sa = a.scan.sin_a[b], r = a.scan.range[b];
        const double c = cs * ca - sn * sa;
        const double s = sn * ca + cs * sa;
        const double wx = x + r * c;
        const double wy = y + r * s;
        const double pr = point_probability<MODEL>(a.map, a.oie, to_cell(wx, scale, inv_scale), to_cell(wy, scale, inv_scale));
        const double term = pr * a.scan.weight[b] * a.scan.factor[b];
        if (WRITE_TERMS) a.terms[(size_t)(p0 + j) * n + b] = term;
        acc = acc + term;
        if (FPRINT) {
          const unsigned kq = (unsigned)(b / kBlock);
          h += term_fingerprint(term, fk_lo + kq * (2u * kBlock * 0x9E3779B1u), fk_hi + kq * (2u * kBlock * 0x85EBCA6Bu));
        }
      }
    }
    if (FPRINT) {
      wave_xor_sum_with(acc, h);
      if (lane == 0) s_hpart[j][wave] = h;
    } else {
      acc = wave_xor_sum(acc);
    }
    if (lane == 0) s_part[j][wave] = acc;
  }
  __syncthreads();
  if (t < npb) {
    const double total = (s_part[t][0] + s_part[t][1]) + (s_part[t][2] + s_part[t][3]);
    a.scores[p0 + t] = (a.scan.tot_w == 0.0) ? __builtin_nan("") : total / a.scan.tot_w;
    if (FPRINT) a.fprints[p0 + t] = s_hpart[t][0] + s_hpart[t][1] + s_hpart[t][2] + s_hpart[t][3];
  }
}

// ---- K1s: the reference's sequential beam-order sum (bit-exact), one lane per pose -------------
__global__ __launch_bounds__(64) void k_sum_sequential(const double *terms, int n_poses, int n,
                                                      double tot_w, double *scores) {
  const int p = blockIdx.x * 64 + threadIdx.x;
  if (p < n_poses) {
    const double *row = terms + (size_t)p * n;
    double acc = 0.0;
    for (int b = 0; b < n; ++b) acc = acc + row[b];
    scores[p] = (tot_w == 0.0) ? __builtin_nan("") : acc / tot_w;
  }
}

// ---- K3: GMapping OOPE (gm_fresh_value and the one-pose body of the wide kernel: gm_score_device.h) ------
// Run-cache quirk (Q19): in beam order every maximal run of equal endpoint cells takes the value
// computed for the run's first beam.  Thread t owns beams t + 256k, so for a fixed k one wave
// holds 64 CONSECUTIVE beams (group g = 4k + wave): run heads come from a ballot + clz inside
// the group and a short backward walk over per-group summaries in LDS across groups.
// ONE: the workgroup scores exactly one pose (every launch below 1024 poses, i.e. all matcher and
// filter launches): without the pose loop the beam constants die after phase A / C instead of
// staying live for a next iteration -- 163 -> fewer VGPRs, more waves to hide the 9-cell gathers.
template <int KB, bool ONE>
__global__ __launch_bounds__(kBlock, (ONE && KB <= 5) ? 5 : 1) void k_score_gmapping(ScoreArgs a) {
  extern __shared__ double s_dyn[];  // val[n] | grp_last_cell (int2 as double) [G] | grp_last_start [G]
  __shared__ double s_pose[kMaxPosesPerBlock][4];
  __shared__ double s_part[kMaxPosesPerBlock][4];
  __shared__ double s_unknown[4];
  __shared__ int s_run0_len;
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int n = a.scan.n;
  const int G = (n + 63) >> 6;
  double *s_val = s_dyn;
  int2 *s_grp_cell = reinterpret_cast<int2 *>(s_dyn + (size_t)KB * kBlock);
  int *s_grp_start = reinterpret_cast<int *>(s_grp_cell + 4 * KB);
  if (t == 0) {  // before the first barrier below
    s_unknown[0] = a.map.unknown[0];
    s_unknown[1] = a.map.unknown[1];
    s_unknown[2] = a.map.unknown[2];
  }
  // Per-particle maps: workgroups go to the 8 XCDs round-robin by index, and consecutive poses read the
  // same particle's tiles.  Handing XCD x the x-th CONTIGUOUS eighth of the poses keeps a particle's
  // tiles in one L2 instead of all eight (the launch rounds the grid up to a multiple of 8).
  int vb = (int)blockIdx.x;
  if (a.xcd_blocks) {
    vb = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
    if (vb >= a.xcd_blocks) return;
  }
  const int p0 = ONE ? vb : vb * a.poses_per_block;
  const int npb = ONE ? 1 : min(a.poses_per_block, a.n_poses - p0);

  // pose loads (PCIe) first, the scan constants' right behind them: see k_score_point
  double pose_x = 0.0, pose_y = 0.0, pose_th = 0.0, pose_sn = 0.0, pose_cs = 0.0;
  if (t < npb) {
    const int p = p0 + t;
    pose_th = a.poses[3 * p + 2];
    pose_x = a.poses[3 * p];
    pose_y = a.poses[3 * p + 1];
    if (a.pose_sc) {
      pose_sn = a.pose_sc[2 * p];
      pose_cs = a.pose_sc[2 * p + 1];
    }
  }
  constexpr int KR = ONE ? 1 : KB;  // (one pose: the beam constants are read where phase A uses them)
  double br[KR], bc[KR], bs[KR], bw[KB], bf[KB];
#pragma unroll
  for (int k = 0; k < KB; ++k) {
    const int b = t + kBlock * k;
    const bool ok = b < n;
    if (!ONE) {
      br[k] = ok ? a.scan.range[b] : 0.0;
      bc[k] = ok ? a.scan.cos_a[b] : 0.0;
      bs[k] = ok ? a.scan.sin_a[b] : 0.0;
    }
    // one pose: weight and factor are read where they are used (phase C) instead of occupying
    // 4 KB of VGPRs across the gathers
    bw[k] = (ok && !ONE) ? a.scan.weight[b] : 0.0;
    bf[k] = (ok && !ONE) ? a.scan.factor[b] : 0.0;
  }
  if (t < npb) {
    if (!a.pose_sc) sincos(pose_th, &pose_sn, &pose_cs);
    s_pose[t][0] = pose_x;
    s_pose[t][1] = pose_y;
    s_pose[t][2] = pose_sn;
    s_pose[t][3] = pose_cs;
  }
  __syncthreads();

  const double scale = a.map.scale, inv_scale = a.map.inv_scale;
  for (int j = 0; j < npb; ++j) {
    const double x = s_pose[j][0], y = s_pose[j][1], sn = s_pose[j][2], cs = s_pose[j][3];
    const int *tiles = a.tables ? a.tables + (size_t)a.pose_slot[p0 + j] * a.table_stride : nullptr;
    int ccx[KB], ccy[KB];
    if (t == 0) s_run0_len = n;
    // phase A: endpoint cell + fresh value per beam
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      const int b = t + kBlock * k;
      ccx[k] = 0;
      ccy[k] = 0;
      if (b < n) {
        const double rk = ONE ? a.scan.range[b] : br[ONE ? 0 : k];
        const double ck = ONE ? a.scan.cos_a[b] : bc[ONE ? 0 : k], sk = ONE ? a.scan.sin_a[b] : bs[ONE ? 0 : k];
        const double c = cs * ck - sn * sk;
        const double s = sn * ck + cs * sk;
        const double wx = x + rk * c;
        const double wy = y + rk * s;
        ccx[k] = to_cell(wx, scale, inv_scale);
        ccy[k] = to_cell(wy, scale, inv_scale);
        s_val[b] = gm_fresh_value(a.map, s_unknown, tiles, a.gm, ccx[k], ccy[k], wx, wy);
        if (lane == 63 || b == n - 1) s_grp_cell[4 * k + wave] = make_int2(ccx[k], ccy[k]);
      }
    }
    __syncthreads();
    // phase B: run starts
    unsigned long long mask[KB];
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      const int b = t + kBlock * k;
      const int g = 4 * k + wave;
      int pcx = __shfl_up(ccx[k], 1, 64), pcy = __shfl_up(ccy[k], 1, 64);
      if (lane == 0 && g > 0 && b < n) {
        const int2 pc = s_grp_cell[g - 1];
        pcx = pc.x;
        pcy = pc.y;
      }
      const bool start = (b < n) && (b == 0 || pcx != ccx[k] || pcy != ccy[k]);
      mask[k] = __ballot(start);
      if (lane == 0 && g < G) s_grp_start[g] = mask[k] ? (64 * g + 63 - __clzll(mask[k])) : -1;
      // first run start behind beam 0: one LDS atomic per wave, from the ballot (one per starting beam --
      // in a real scan nearly every beam -- queued a thousand lanes on the one word: 8 of a lone
      // launch's 21 us)
      const unsigned long long later = g == 0 ? mask[k] & ~1ull : mask[k];
      if (lane == 0 && later) atomicMin(&s_run0_len, 64 * g + __ffsll((long long)later) - 1);
    }
    __syncthreads();
    // phase C: resolve run heads, accumulate in the canonical order
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      const int b = t + kBlock * k;
      if (b < n) {
        const int g = 4 * k + wave;
        const unsigned long long upto = mask[k] & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
        int head;
        if (upto) {
          head = 64 * g + 63 - __clzll(upto);
        } else {
          int gg = g - 1;
          head = s_grp_start[gg];
          while (head < 0) head = s_grp_start[--gg];  // beam 0 is always a start
        }
        const double v = s_val[head];
        const double wk = ONE ? a.scan.weight[b] : bw[k], fk = ONE ? a.scan.factor[b] : bf[k];
        const double term = v * wk * fk;
        acc = acc + term;
        if (b == n - 1 && a.gm_info) {
          GmPoseInfo &gi = a.gm_info[p0 + j];
          gi.last_cx = ccx[k];
          gi.last_cy = ccy[k];
          gi.last_v = v;
          gi.last_head = head;
        }
        if (b == 0 && a.gm_info) {
          GmPoseInfo &gi = a.gm_info[p0 + j];
          gi.first_cx = ccx[k];
          gi.first_cy = ccy[k];
          gi.v0 = v;
        }
      }
    }
    acc = wave_xor_sum(acc);
    if (lane == 0) s_part[j][wave] = acc;
    __syncthreads();
    if (t == 0 && a.gm_info) a.gm_info[p0 + j].run0_len = s_run0_len;
  }
  __syncthreads();
  if (t < npb) {
    const double total = (s_part[t][0] + s_part[t][1]) + (s_part[t][2] + s_part[t][3]);
    a.scores[p0 + t] = (a.scan.tot_w == 0.0) ? __builtin_nan("") : total / a.scan.tot_w;
  }
}

// K3, one pose per workgroup of 1024 threads, for launches of a few dozen poses (a lone matcher's batches
// leave the GPU empty, so the per-pose chain of phases is the whole kernel time): the expensive phase A
// (end point, nine gathers, exp) is spread over sixteen waves -- beam b goes to thread b % 1024 -- while the
// run resolution and the sum keep the canonical 256-thread layout (thread t < 256 owns beams t + 256k, read
// back from LDS), so scores are bit-identical to k_score_gmapping's.  (A 512-thread form for the filter's
// launches of a few hundred poses measured equal -- 41.5 against 41.1 us -- and was removed.)
template <int KB, int NT>
__global__ __launch_bounds__(NT) void k_score_gmapping_wide(ScoreArgs a) {
  extern __shared__ double s_dyn[];  // val[256 KB] | grp_cell int2 [4 KB] | grp_start int [4 KB] | cx, cy int [256 KB]
  __shared__ double s_pose1[4];
  __shared__ double s_part1[4];
  __shared__ double s_unknown[4];
  __shared__ int s_run0_len;
  const int t = threadIdx.x;
  const int n = a.scan.n;
  const int p = blockIdx.x;
  // pose loads (PCIe) first, every thread's first beam right behind them: see k_score_point
  double pose_x = 0.0, pose_y = 0.0, pose_th = 0.0, pose_sn = 0.0, pose_cs = 0.0;
  if (t == 0) {
    pose_th = a.poses[3 * p + 2];
    pose_x = a.poses[3 * p];
    pose_y = a.poses[3 * p + 1];
    if (a.pose_sc) {
      pose_sn = a.pose_sc[2 * p];
      pose_cs = a.pose_sc[2 * p + 1];
    }
  }
  const bool has0 = t < n;
  const double r0 = has0 ? a.scan.range[t] : 0.0, ca0 = has0 ? a.scan.cos_a[t] : 0.0, sa0 = has0 ? a.scan.sin_a[t] : 0.0;
  if (t == 64) {
    s_unknown[0] = a.map.unknown[0];
    s_unknown[1] = a.map.unknown[1];
    s_unknown[2] = a.map.unknown[2];
  }
  if (t == 0) {
    if (!a.pose_sc) sincos(pose_th, &pose_sn, &pose_cs);
    s_pose1[0] = pose_x;
    s_pose1[1] = pose_y;
    s_pose1[2] = pose_sn;
    s_pose1[3] = pose_cs;
    s_run0_len = n;
  }
  __syncthreads();
  const int *tiles = a.tables ? a.tables + (size_t)a.pose_slot[p] * a.table_stride : nullptr;
  double score;
  gm_score_pose_wide<KB, NT>(a.map, a.scan, a.gm, tiles, s_unknown, s_pose1[0], s_pose1[1], s_pose1[2], s_pose1[3], r0, ca0,
                             sa0, s_dyn, &s_run0_len, s_part1, a.gm_info ? a.gm_info + p : nullptr, &score);
  if (t == 0) a.scores[p] = score;
}

// (the window OOPEs' per-beam value: window_probability, score_device.h -- shared with the co-resident chain)
template <int MODEL>
__global__ __launch_bounds__(kBlock) void k_score_window(ScoreArgs a, int oope) {
  __shared__ double s_pose[kMaxPosesPerBlock][4];
  __shared__ double s_part[kMaxPosesPerBlock][4];
  __shared__ unsigned long long s_hpart[kMaxPosesPerBlock][4];
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int n = a.scan.n;
  const int p0 = blockIdx.x * a.poses_per_block;
  const int npb = min(a.poses_per_block, a.n_poses - p0);
  // (the checked default mode over the window OOPEs, r05: K1's term-vector fingerprint, same multipliers)
  const bool fprint = a.fprints != nullptr;
  const unsigned fk_lo = (2u * (unsigned)t + 1u) * 0x9E3779B1u, fk_hi = (2u * (unsigned)t + 1u) * 0x85EBCA6Bu;
  if (t < npb) {
    const int p = p0 + t;
    double sn, cs;
    if (a.pose_sc) {
      sn = a.pose_sc[2 * p];
      cs = a.pose_sc[2 * p + 1];
    } else {
      sincos(a.poses[3 * p + 2], &sn, &cs);
    }
    s_pose[t][0] = a.poses[3 * p];
    s_pose[t][1] = a.poses[3 * p + 1];
    s_pose[t][2] = sn;
    s_pose[t][3] = cs;
  }
  __syncthreads();
  const double half_v = (a.area[1] - a.area[0]) / 2, half_h = (a.area[3] - a.area[2]) / 2;
  for (int j = 0; j < npb; ++j) {
    const double x = s_pose[j][0], y = s_pose[j][1], sn = s_pose[j][2], cs = s_pose[j][3];
    double acc = 0.0;
    unsigned long long h = 0ull;
    for (int b = t; b < n; b += kBlock) {
      const double ca = a.scan.cos_a[b], sa = a.scan.sin_a[b], r = a.scan.range[b];
      const double c = cs * ca - sn * sa;
      const double s = sn * ca + cs * sa;
      const double ox = x + r * c, oy = y + r * s;
      const double pr = window_probability<MODEL>(a.map, a.oie, oope, half_v, half_h, ox, oy);
      const double term = pr * a.scan.weight[b] * a.scan.factor[b];
      if (a.terms) a.terms[(size_t)(p0 + j) * n + b] = term;
      acc = acc + term;
      if (fprint) {
        const unsigned kq = (unsigned)(b / kBlock);
        h += term_fingerprint(term, fk_lo + kq * (2u * kBlock * 0x9E3779B1u), fk_hi + kq * (2u * kBlock * 0x85EBCA6Bu));
      }
    }
    wave_xor_sum_with(acc, h);  // (the butterfly of wave_xor_sum with the fingerprint riding along: same sum bits)
    if (lane == 0) {
      s_part[j][wave] = acc;
      s_hpart[j][wave] = h;
    }
  }
  __syncthreads();
  if (t < npb) {
    const double total = (s_part[t][0] + s_part[t][1]) + (s_part[t][2] + s_part[t][3]);
    a.scores[p0 + t] = (a.scan.tot_w == 0.0) ? __builtin_nan("") : total / a.scan.tot_w;
    if (fprint) a.fprints[p0 + t] = s_hpart[t][0] + s_hpart[t][1] + s_hpart[t][2] + s_hpart[t][3];
  }
}

// ---- launch ------------------------------------------------------------------------------------
// Scoring launches go through hipExtLaunchKernelGGL so that slamhip_profile_* can attach its
// HIP events to the dispatch itself: the elapsed time is then the kernel's own begin..end (what
// rocprofv3 --kernel-trace reports), not record-to-record on an idle stream which adds ~4 us of
// queue processing per isolated launch (tools/event_probe.hip).  Without events the ordinary launch
// is used.
#define SLAMHIP_LAUNCH(kernel, grid, block, shm, st, e0, e1, ...)                          \
  do {                                                                                     \
    if ((e0) || (e1))                                                                      \
      hipExtLaunchKernelGGL(kernel, grid, block, shm, st, e0, e1, 0, __VA_ARGS__);         \
    else                                                                                   \
      hipLaunchKernelGGL(kernel, grid, block, shm, st, __VA_ARGS__);                       \
  } while (0)

template <int MODEL, bool WT, bool FP = false>
static hipError_t launch_point_kb(const ScoreArgs &a, int kb, dim3 grid, hipStream_t st,
                                  hipEvent_t e0, hipEvent_t e1) {
  switch (kb) {
    case 1: SLAMHIP_LAUNCH((k_score_point<MODEL, 1, WT, FP>), grid, dim3(kBlock), 0, st, e0, e1, a); break;
    case 2: SLAMHIP_LAUNCH((k_score_point<MODEL, 2, WT, FP>), grid, dim3(kBlock), 0, st, e0, e1, a); break;
    case 3: SLAMHIP_LAUNCH((k_score_point<MODEL, 3, WT, FP>), grid, dim3(kBlock), 0, st, e0, e1, a); break;
    case 4: SLAMHIP_LAUNCH((k_score_point<MODEL, 4, WT, FP>), grid, dim3(kBlock), 0, st, e0, e1, a); break;
    case 5: SLAMHIP_LAUNCH((k_score_point<MODEL, 5, WT, FP>), grid, dim3(kBlock), 0, st, e0, e1, a); break;
    default: SLAMHIP_LAUNCH((k_score_point<MODEL, 0, WT, FP>), grid, dim3(kBlock), 0, st, e0, e1, a); break;
  }
  return hipGetLastError();
}

static int pick_poses_per_block(int n_poses) {
  // enough workgroups to fill 256 CUs several times over, while amortising the per-workgroup
  // beam-constant loads (40 B/beam) over several poses once there are plenty of poses.
  if (n_poses >= 8192) return 8;
  if (n_poses >= 2048) return 4;
  if (n_poses >= 1024) return 2;
  return 1;
}

hipError_t launch_score(const ScoreArgs &args, int cell_model, int oope, int sum_order,
                        hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop) {
  ScoreArgs a = args;
  if (a.n_poses <= 0) return hipSuccess;
  if (a.poses_per_block <= 0) {
    a.poses_per_block = pick_poses_per_block(a.n_poses);
    // K3 up to a couple of thousand poses does not fill the chip's wave slots, and the poses of a
    // workgroup run one after the other: one pose per workgroup (100-particle filter step 1.69 -> 1.66 ms)
    constexpr int gm_one_below = 2048;
    // (through tile tables at any size: the multi-pose body's 163 VGPRs leave three waves per SIMD to hide the
    // nine gathers per beam -- 3000 poses of a 500-particle round: 262 -> 136 us)
    if (oope == SLAMHIP_OOPE_GMAPPING && (a.n_poses < gm_one_below || a.tables)) a.poses_per_block = 1;
  }
  if (a.poses_per_block > kMaxPosesPerBlock) a.poses_per_block = kMaxPosesPerBlock;
  const dim3 grid((a.n_poses + a.poses_per_block - 1) / a.poses_per_block);
  const int kb = (a.scan.n + kBlock - 1) / kBlock;
  const bool wt = sum_order == SLAMHIP_SUM_SEQUENTIAL;
  constexpr bool xcd_off = false;
  a.xcd_blocks = 0;
  if (wt && oope != SLAMHIP_OOPE_GMAPPING && (ev_start || ev_stop)) return hipErrorInvalidValue;
  const hipEvent_t stop1 = ev_stop;
  hipError_t e = hipSuccess;
  if (oope == SLAMHIP_OOPE_GMAPPING) {
    if (kb > 8) {
      set_error("the GMapping kernel holds at most 2048 filtered beams per scan");
      return hipErrorInvalidValue;
    }
    const size_t shm = (size_t)kb * kBlock * sizeof(double) + 4 * kb * sizeof(int2) + 4 * kb * sizeof(int);
    // 1024 threads per pose for launches of at most 160 poses
    constexpr int wide_below = 160;
    const int wide = a.n_poses <= wide_below ? 1024 : 0;
    const size_t shm_wide = shm + 2 * (size_t)kb * kBlock * sizeof(int) + kGmHelperDoubles<1024> * sizeof(double);
    dim3 grid_gm = grid;  // k_score_gmapping only: the XCD-chunked block order (see the kernel)
    if (a.tables && !xcd_off && grid.x >= 16 && !(a.poses_per_block == 1 && wide)) {
      a.xcd_blocks = (int)grid.x;
      grid_gm.x = (grid.x + 7) / 8 * 8;
    }
#define GM_CASE(K)                                                                                            \
  case K:                                                                                                     \
    if (a.poses_per_block == 1 && wide == 1024)                                                               \
      SLAMHIP_LAUNCH((k_score_gmapping_wide<K, 1024>), grid, dim3(1024), shm_wide, stream, ev_start, ev_stop, a); \
    else if (a.poses_per_block == 1)                                                                          \
      SLAMHIP_LAUNCH((k_score_gmapping<K, true>), grid_gm, dim3(kBlock), shm, stream, ev_start, ev_stop, a);  \
    else                                                                                                      \
      SLAMHIP_LAUNCH((k_score_gmapping<K, false>), grid_gm, dim3(kBlock), shm, stream, ev_start, ev_stop, a); \
    break;
    switch (kb < 1 ? 1 : kb) {
      GM_CASE(1) GM_CASE(2) GM_CASE(3) GM_CASE(4) GM_CASE(5) GM_CASE(6) GM_CASE(7) GM_CASE(8)
    }
#undef GM_CASE
    return hipGetLastError();
  }
  if (oope == SLAMHIP_OOPE_MAX || oope == SLAMHIP_OOPE_MEAN || oope == SLAMHIP_OOPE_OVERLAP) {
    if (!wt) a.terms = nullptr;
    if (wt) a.fprints = nullptr;
    if (cell_model == SLAMHIP_CELL_OCC)
      SLAMHIP_LAUNCH((k_score_window<SLAMHIP_CELL_OCC>), grid, dim3(kBlock), 0, stream, ev_start, stop1, a, oope);
    else if (cell_model == SLAMHIP_CELL_TBM)
      SLAMHIP_LAUNCH((k_score_window<SLAMHIP_CELL_TBM>), grid, dim3(kBlock), 0, stream, ev_start, stop1, a, oope);
    else
      return hipErrorInvalidValue;
    e = hipGetLastError();
  } else if (cell_model == SLAMHIP_CELL_OCC) {
    e = wt ? launch_point_kb<SLAMHIP_CELL_OCC, true>(a, kb, grid, stream, ev_start, stop1)
           : (a.fprints ? launch_point_kb<SLAMHIP_CELL_OCC, false, true>(a, kb, grid, stream, ev_start, stop1)
                        : launch_point_kb<SLAMHIP_CELL_OCC, false>(a, kb, grid, stream, ev_start, stop1));
  } else if (cell_model == SLAMHIP_CELL_TBM) {
    e = wt ? launch_point_kb<SLAMHIP_CELL_TBM, true>(a, kb, grid, stream, ev_start, stop1)
           : (a.fprints ? launch_point_kb<SLAMHIP_CELL_TBM, false, true>(a, kb, grid, stream, ev_start, stop1)
                        : launch_point_kb<SLAMHIP_CELL_TBM, false>(a, kb, grid, stream, ev_start, stop1));
  } else {
    return hipErrorInvalidValue;
  }
  if (e != hipSuccess) return e;
  if (wt) {  // strict order is two kernels: the caller times the pair with recorded events instead
    SLAMHIP_LAUNCH(k_sum_sequential, dim3((a.n_poses + 63) / 64), dim3(64), 0, stream, (hipEvent_t) nullptr,
                   (hipEvent_t) nullptr, a.terms, a.n_poses, a.scan.n, a.scan.tot_w, a.scores);
    e = hipGetLastError();
  }
  return e;
}

// ---- map mirror maintenance --------------------------------------------------------------------
__global__ void k_fill_cells(double *dst, size_t n_cells, int cell_dbl, double u0, double u1,
                             double u2, double u3) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t c = i; c < n_cells; c += stride) {
    if (cell_dbl == 1) {
      dst[c] = u0;
    } else {
      reinterpret_cast<double4 *>(dst)[c] = make_double4(u0, u1, u2, u3);
    }
  }
}

__global__ void k_repack_window(double *dst, int dst_pitch, int cell_dbl, const double *src,
                                int stride_host, int x0, int y0, int w, int h) {
  const size_t total = (size_t)w * h;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int yy = (int)(i / w), xx = (int)(i % w);
    double *d = dst + ((size_t)(y0 + yy) * dst_pitch + (x0 + xx)) * cell_dbl;
    const double *s = src + i * stride_host;
    for (int k = 0; k < stride_host && k < cell_dbl; ++k) d[k] = s[k];
  }
}

__global__ void k_scatter_cells(double *payload, int pitch, int cell_dbl, int stride_host, int n,
                                const int *coords, const double *vals) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double *d = payload + ((size_t)coords[2 * i + 1] * pitch + coords[2 * i]) * cell_dbl;
  for (int k = 0; k < stride_host && k < cell_dbl; ++k) d[k] = vals[(size_t)i * stride_host + k];
}

// ---- neighbourhood masks of a dense GMAPPING window (MapView) ------------------------------------------------
__device__ __forceinline__ unsigned nbr_mask_of(const double *payload, int width, int height, int pitch, double th, int ix,
                                                int iy) {
  unsigned m9 = 0u;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int x = ix + i / 3 - 1, y = iy + i % 3 - 1;
    if ((unsigned)x < (unsigned)width && (unsigned)y < (unsigned)height && !(payload[4 * ((size_t)y * pitch + x)] < th))
      m9 |= 1u << i;
  }
  return m9;
}

__global__ __launch_bounds__(256) void k_nbr_build(double *payload, int width, int height, int pitch, double th, int x0, int y0,
                                                   int w, int h) {
  const size_t total = (size_t)w * h;
  for (size_t c = (size_t)blockIdx.x * 256 + threadIdx.x; c < total; c += (size_t)gridDim.x * 256) {
    const int iy = y0 + (int)(c / w), ix = x0 + (int)(c % w);
    reinterpret_cast<unsigned *>(payload + 4 * ((size_t)iy * pitch + ix) + 3)[0] =
        nbr_mask_of(payload, width, height, pitch, th, ix, iy);
  }
}

__global__ __launch_bounds__(256) void k_nbr_cells(double *payload, int width, int height, int pitch, double th, int n,
                                                   const int *coords) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= 9 * n) return;
  const int c = j / 9, i = j - 9 * c;
  const int ix = coords[2 * c] + i / 3 - 1, iy = coords[2 * c + 1] + i % 3 - 1;
  if ((unsigned)ix >= (unsigned)width || (unsigned)iy >= (unsigned)height) return;
  // (two listed cells next to each other re-derive the same mask twice: the same value both times)
  reinterpret_cast<unsigned *>(payload + 4 * ((size_t)iy * pitch + ix) + 3)[0] =
      nbr_mask_of(payload, width, height, pitch, th, ix, iy);
}

__global__ __launch_bounds__(256) void k_nbr_check(const double *payload, int width, int height, int pitch, double th,
                                                   unsigned long long *count) {
  const size_t total = (size_t)width * height;
  unsigned long long bad = 0;
  for (size_t c = (size_t)blockIdx.x * 256 + threadIdx.x; c < total; c += (size_t)gridDim.x * 256) {
    const int iy = (int)(c / width), ix = (int)(c % width);
    const unsigned have = reinterpret_cast<const unsigned *>(payload + 4 * ((size_t)iy * pitch + ix) + 3)[0];
    if (have != nbr_mask_of(payload, width, height, pitch, th, ix, iy)) ++bad;
  }
  if (bad) atomicAdd(count, bad);
}

hipError_t launch_nbr_build(double *payload, int width, int height, int pitch, double th, int x0, int y0, int w, int h,
                            hipStream_t stream) {
  const int x1 = std::min(width, x0 + w), y1 = std::min(height, y0 + h);
  x0 = std::max(0, x0);
  y0 = std::max(0, y0);
  if (x1 <= x0 || y1 <= y0) return hipSuccess;
  const size_t total = (size_t)(x1 - x0) * (y1 - y0);
  const int blocks = (int)std::min<size_t>((total + 255) / 256, (size_t)65536);
  hipLaunchKernelGGL(k_nbr_build, dim3(blocks), dim3(256), 0, stream, payload, width, height, pitch, th, x0, y0, x1 - x0,
                     y1 - y0);
  return hipGetLastError();
}

hipError_t launch_nbr_cells(double *payload, int width, int height, int pitch, double th, int n, const int *d_coords,
                            hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_nbr_cells, dim3((9 * n + 255) / 256), dim3(256), 0, stream, payload, width, height, pitch, th, n,
                     d_coords);
  return hipGetLastError();
}

hipError_t launch_nbr_check(const double *payload, int width, int height, int pitch, double th, unsigned long long *d_count,
                            hipStream_t stream) {
  const size_t total = (size_t)width * height;
  const int blocks = (int)std::min<size_t>((total + 255) / 256, (size_t)65536);
  hipLaunchKernelGGL(k_nbr_check, dim3(blocks ? blocks : 1), dim3(256), 0, stream, payload, width, height, pitch, th, d_count);
  return hipGetLastError();
}

// ---- the probability plane of a TBM map (DeviceMap::d_prob) ---------------------------------------------------
__global__ __launch_bounds__(256) void k_prob_build(const double4 *__restrict__ cells, double *__restrict__ prob, int pitch, int x0,
                                                   int y0, int w, int h) {
  const int x = x0 + (int)(blockIdx.x * 256 + threadIdx.x), y = y0 + (int)blockIdx.y;
  if (x >= x0 + w || y >= y0 + h) return;
  const size_t at = (size_t)y * pitch + x;
  const double4 v = cells[at];
  prob[at] = tbm_discrepancy_probability(v.x, v.y, v.z, v.w);
}
__global__ __launch_bounds__(256) void k_prob_cells(const double4 *__restrict__ cells, double *__restrict__ prob, int width, int height,
                                                   int pitch, int n, const int *__restrict__ coords) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int x = coords[2 * i], y = coords[2 * i + 1];
  if ((unsigned)x >= (unsigned)width || (unsigned)y >= (unsigned)height) return;
  const size_t at = (size_t)y * pitch + x;
  const double4 v = cells[at];
  prob[at] = tbm_discrepancy_probability(v.x, v.y, v.z, v.w);
}
__global__ __launch_bounds__(256) void k_prob_check(const double4 *__restrict__ cells, const double *__restrict__ prob, int width,
                                                   int pitch, unsigned long long *count) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= width) return;
  const size_t at = (size_t)y * pitch + x;
  const double4 v = cells[at];
  const double want = tbm_discrepancy_probability(v.x, v.y, v.z, v.w);
  if (__double_as_longlong(want) != __double_as_longlong(prob[at])) atomicAdd(count, 1ull);
}
hipError_t launch_prob_build(const double *payload, double *prob, int width, int height, int pitch, int x0, int y0, int w, int h,
                             hipStream_t stream) {
  const int xa = x0 < 0 ? 0 : x0, ya = y0 < 0 ? 0 : y0;
  const int xb = x0 + w > width ? width : x0 + w, yb = y0 + h > height ? height : y0 + h;
  if (xb <= xa || yb <= ya) return hipSuccess;
  hipLaunchKernelGGL(k_prob_build, dim3((xb - xa + 255) / 256, yb - ya), dim3(256), 0, stream,
                     reinterpret_cast<const double4 *>(payload), prob, pitch, xa, ya, xb - xa, yb - ya);
  return hipGetLastError();
}
hipError_t launch_prob_cells(const double *payload, double *prob, int width, int height, int pitch, int n, const int *d_coords,
                             hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_prob_cells, dim3((n + 255) / 256), dim3(256), 0, stream, reinterpret_cast<const double4 *>(payload), prob,
                     width, height, pitch, n, d_coords);
  return hipGetLastError();
}
hipError_t launch_prob_check(const double *payload, const double *prob, int width, int height, int pitch,
                             unsigned long long *d_count, hipStream_t stream) {
  hipLaunchKernelGGL(k_prob_check, dim3((width + 255) / 256, height), dim3(256), 0, stream,
                     reinterpret_cast<const double4 *>(payload), prob, width, pitch, d_count);
  return hipGetLastError();
}

hipError_t launch_fill_cells(double *dst, size_t n_cells, int cell_dbl, const double *u,
                             hipStream_t stream) {
  const int blocks = (int)std::min<size_t>((n_cells + 255) / 256, (size_t)4096);
  hipLaunchKernelGGL(k_fill_cells, dim3(blocks ? blocks : 1), dim3(256), 0, stream, dst, n_cells,
                     cell_dbl, u[0], u[1], u[2], u[3]);
  return hipGetLastError();
}

hipError_t launch_repack_window(double *dst, int dst_pitch, int cell_dbl, const double *src,
                                int stride_host, int x0, int y0, int w, int h, hipStream_t stream) {
  const size_t total = (size_t)w * h;
  const int blocks = (int)std::min<size_t>((total + 255) / 256, (size_t)4096);
  hipLaunchKernelGGL(k_repack_window, dim3(blocks ? blocks : 1), dim3(256), 0, stream, dst, dst_pitch,
                     cell_dbl, src, stride_host, x0, y0, w, h);
  return hipGetLastError();
}

hipError_t launch_scatter_cells(double *payload, int pitch, int cell_dbl, int stride_host, int n,
                                const int *d_coords, const double *d_vals, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_scatter_cells, dim3((n + 255) / 256), dim3(256), 0, stream, payload, pitch,
                     cell_dbl, stride_host, n, d_coords, d_vals);
  return hipGetLastError();
}

}  // namespace slamhip
