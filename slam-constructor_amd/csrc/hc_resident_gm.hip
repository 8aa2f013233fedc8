// hc_resident_gm.hip -- the co-resident hill-climbing chain (hc_resident.hip) over the GMapping OOPE.
//
//   GmappingOccupancyObservationPE (+ its cross-pose cache)   src/slams/gmapping/gmapping_occupancy_observation_pe.h:17-44
//   PoseEnumerationScanMatcher::process_scan                  src/core/scan_matchers/pose_enumeration_scan_matcher.h:31-77
//   GmappingWorld::handle_observation (one match per particle) src/slams/gmapping/gmapping_world.h:73-101
//
// Same structure as the 1-cell form: 6 x instances + 1 one-pose workgroups launched once per match (or once for all
// particles of a filter step, grid.y = chain), scores exchanged inside the launch, every workgroup replays.  What a
// pose hands to the replay is more than its score here: the replay applies the reference's cross-pose cache (Q19)
// to the walked path in call order (hc_gm_fix), which needs every pose's side outputs (GmPoseInfo: first / last end
// cell, the first run's fresh value and length, the last run's value and head).  A slot therefore publishes FOUR
// 16-byte granules, each {12 bytes of payload, tag}, in one store instruction of four lanes; waves 0..3 sweep one
// granule kind each into LDS, a workgroup barrier, then wave 0 replays -- hc_chain.hip's GMapping replay, line by
// line.  Scoring is K3's one-pose body (gm_score_pose_wide): same bits as every other path.
// Bounded waits, fail-over and tags: hc_resident.hip / hc_resident_device.h.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "gm_score_device.h"
#include "hc_chain_device.h"
#include "hc_resident_device.h"
#include "score_device.h"

namespace slamhip {

namespace {

// MatchJob's gm_apply_carry (matchers.h) for one pose: what the cross-pose cache does to a replayed pose's score,
// and what the pose leaves in it (hc_chain.hip)
struct HcCarry {
  int cx, cy;
  double prob;
};
__device__ __forceinline__ double hc_gm_fix(double score, HcCarry &cr, const GmPoseInfo &gi, const ScanView &scan) {
  double last_v = gi.last_v;
  if (cr.prob != -1.0 && gi.first_cx == cr.cx && gi.first_cy == cr.cy) {
    const double c = cr.prob;
    if (c != gi.v0) {
      double delta = 0.0;
      for (int b = 0; b < gi.run0_len; ++b)
        delta += (c * scan.weight[b]) * scan.factor[b] - (gi.v0 * scan.weight[b]) * scan.factor[b];
      if (scan.tot_w != 0.0) score += delta / scan.tot_w;
    }
    if (gi.last_head == 0) last_v = c;
  }
  cr.cx = gi.last_cx;
  cr.cy = gi.last_cy;
  cr.prob = last_v;
  return score;
}

// granule kind q of a pose: {8 bytes, 4 bytes} of payload (the tag rides in the last dword)
__device__ __forceinline__ void gm_gran_store(HcGranule *p, int q, double score, const GmPoseInfo &gi, unsigned tag) {
  double d = 0.0;
  unsigned lo = 0u, hi = 0u;
  switch (q) {
    case 0: d = score; lo = (unsigned)gi.run0_len; break;
    case 1: d = gi.v0; lo = (unsigned)gi.last_head; break;
    case 2: d = gi.last_v; lo = (unsigned)gi.first_cx; break;
    default: {
      const unsigned long long w = ((unsigned long long)(unsigned)gi.last_cx << 32) | (unsigned)gi.first_cy;
      d = __longlong_as_double((long long)w);
      lo = (unsigned)gi.last_cy;
      break;
    }
  }
  (void)hi;
  const unsigned long long u = (unsigned long long)__double_as_longlong(d);
  u32x4 g;
  g.x = (unsigned)u;
  g.y = (unsigned)(u >> 32);
  g.z = lo;
  g.w = tag & 0xffffu;
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(g) : "memory");
}

}  // namespace

// NT threads per pose (256 / 512 / 1024), KB = ceil(beams / 256), G = sweeping waves, a slot per lane (the grid has
// at most 64 G workgroups: 1 for the filter's many small trees, 4 for a lone chain's 253);
// gran4: [2][kHcSlots + 7][4] granules per chain
// (256-thread workgroups -- the filter's hundred small trees -- get three waves per SIMD, 168 VGPRs: K3's one-pose body
// spills 27 registers at four, and 3 x 256 workgroups per CU still hold the 700 of a 100-particle step)
template <int NT, int KB, int G>
__global__ __launch_bounds__(NT, (NT == 256 ? 3 : 4)) void k_hc_chain_resident_gm(HcChainArgs a) {
  constexpr int kRow = kHcSlots + 7;
  extern __shared__ double s_dyn[];  // K3's arrays (gm_score_pose_wide)
  __shared__ GmPoseInfo s_info[kRow];
  __shared__ double s_sc[kRow];
  __shared__ HcInst s_mine[kHcShapes];
  __shared__ double s_pose[2][4];
  __shared__ int s_go[2];
  __shared__ int s_stop, s_sweep_failed;
  __shared__ HcState s_st;
  __shared__ double s_part[4];
  __shared__ double s_unknown[4];
  __shared__ int s_run0_len;
  __shared__ GmPoseInfo s_gi;   // this workgroup's pose
  __shared__ double s_score;
  __shared__ GmPoseInfo s_first_info;  // (bookkeeping workgroup) side outputs and raw score of the initial pose
  __shared__ double s_first_raw;
  // this thread's first beam (range, cos, sin), read once per match: kept in LDS, not in registers -- K3's one-pose
  // body is at the register limit, and loop-carried registers were spilled to scratch and reloaded (a trip through
  // memory) in front of every pose
  __shared__ double s_beam[3 * NT];
  const int t = threadIdx.x, wave = t >> 6;
  const int slot = blockIdx.x + 1 == gridDim.x ? kHcSlots - 1 : (int)blockIdx.x;
  const int inst_of_slot = slot / 6, cand = slot - 6 * inst_of_slot;
  const bool init_slot = slot == kHcSlots - 1;
  HcResidentGmCtl *const rc = a.rctl_gm + blockIdx.y;
  HcHostOut *const host = a.host + blockIdx.y;
  // (a workgroup that starts after the others gave up leaves at once -- ONE thread's reading, behind the barrier below)
  const unsigned fail_epoch_at_entry = __hip_atomic_load(&rc->fail_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef SLAMHIP_TESTING
  if (a.debug_mute && (int)blockIdx.x + 1 == a.debug_mute) return;  // (the others must give up, not hang)
#endif
  const __attribute__((address_space(4))) HcChainArgs *const ap0 =
      (const __attribute__((address_space(4))) HcChainArgs *)__builtin_amdgcn_kernarg_segment_ptr();
  const ScanView scan = load_view(&ap0->scan);
  const int n = scan.n;
  {
    double br0 = 0.0, bc0 = 0.0, bs0 = 0.0;
    if (t < n) {
      br0 = scan.range[t];
      bc0 = scan.cos_a[t];
      bs0 = scan.sin_a[t];
    }
    s_beam[t] = br0;
    s_beam[NT + t] = bc0;
    s_beam[2 * NT + t] = bs0;
  }
  if (wave == 1 && (t & 63) < kHcShapes && !init_slot) {
    const uint4 *src = reinterpret_cast<const uint4 *>(&a.shapes[t & 63].inst[inst_of_slot]);
    uint4 *dst = reinterpret_cast<uint4 *>(&s_mine[t & 63]);
#pragma unroll
    for (int q = 0; q < (int)(sizeof(HcInst) / 16); ++q) dst[q] = src[q];
  }
  if (t == 64) {
    s_unknown[0] = a.map.unknown[0];
    s_unknown[1] = a.map.unknown[1];
    s_unknown[2] = a.map.unknown[2];
  }
  HcGranule *const gran = &rc->gran[0][0][0];
  if (t < 8) gm_gran_store(gran + ((size_t)(t >> 2) * kRow + slot) * 4 + (t & 3), 0, 0.0, GmPoseInfo{}, 0u);
  if (t == 0) {
    s_stop = fail_epoch_at_entry == a.epoch ? 1 : 0;
    s_sweep_failed = 0;
    HcState st{};
    st.x = a.inits ? a.inits[3 * blockIdx.y] : a.init[0];
    st.y = a.inits ? a.inits[3 * blockIdx.y + 1] : a.init[1];
    st.theta = a.inits ? a.inits[3 * blockIdx.y + 2] : a.init[2];
    st.dt = a.dt0;
    st.dr = a.dr0;
    st.shape = a.shape0;
    st.first = 1;
    st.carry_cx = a.gm_cx;
    st.carry_cy = a.gm_cy;
    st.carry_prob = a.gm_prob;
    s_st = st;
  }
  const int *tiles = a.tables ? a.tables + (size_t)a.slots[blockIdx.y] * a.table_stride : nullptr;
  const bool stamp = SLAMHIP_STAMPS_ON(a.stamps && slot == 1 && t == 0);
  __syncthreads();
  if (s_stop) return;  // (uniform)

  const int t_entry = t;
  for (int k = 0;; ++k) {
    const int pk = k & 1;
    int tt = t_entry;
    asm volatile("" : "+v"(tt));  // (see hc_resident.hip: nothing derived from the thread index is hoisted)
    const int lane = tt & 63;
    // (... and the kernel's arguments and the map view are re-read from the kernarg segment where they are used)
    const __attribute__((address_space(4))) HcChainArgs *ap = ap0;
    asm volatile("" : "+s"(ap));
    const MapView map = load_view(&ap->map);
    GmParams gmp;
    gmp.fullness_th = ap->gm.fullness_th;
    gmp.window = ap->gm.window;
    if (wave == 0) {
      if (stamp && k < 64) ap->stamps[8 * k + 0] = wall_clock64();
      const HcState &st = s_st;
      bool go = !st.done;
      double px = st.x, py = st.y, pth = st.theta;
      if (init_slot) {
        go = go && st.first;  // (no checked mode over the GMapping OOPE: the initial pose only)
      } else if (go) {
        HcInst in;
#pragma unroll
        for (int q = 0; q < 14; ++q) in.w[q] = s_mine[st.shape].w[q];
        go = inst_of_slot < (int)((ap->n_inst >> (8 * st.shape)) & 0xffull) &&
             (hc_is_root(in) || st.failed + hc_nfail_parent(in) < ap->max_failed);
        if (go) {
          const HcRound r = hc_round_of(st, in);
          go = !(hc_trailing(r.failed, ap->max_failed) && cand > 0);
          hc_candidate(r.x, r.y, r.theta, r.dt, r.dr, cand, &px, &py, &pth);
        }
      }
      if (go) {
        double sn, cs;
        sincos(pth, &sn, &cs);
        if (lane == 0) {
          s_pose[pk][0] = px;
          s_pose[pk][1] = py;
          s_pose[pk][2] = sn;
          s_pose[pk][3] = cs;
        }
      }
      if (lane == 0) {
        s_go[pk] = go ? 1 : 0;
        s_run0_len = n;
        if (st.done) s_stop = 1;
      }
      if (stamp && k < 64) ap->stamps[8 * k + 3] = wall_clock64();
    }
    __syncthreads();  // (A)
    if (s_stop) break;
    const int go = s_go[pk];
    const unsigned tag = hc_tag(ap->tag_epoch, k);
    HcGranule *const mine4 = gran + ((size_t)pk * kRow + slot) * 4;
    if (go) {
      const double px = s_pose[pk][0], py = s_pose[pk][1], sn = s_pose[pk][2], cs = s_pose[pk][3];
      double score = 0.0;
      const double br = s_beam[tt], bc = s_beam[NT + tt], bs = s_beam[2 * NT + tt];
      gm_score_pose_wide<KB, NT>(map, scan, gmp, tiles, s_unknown, px, py, sn, cs, br, bc, bs, s_dyn, &s_run0_len, s_part,
                                 &s_gi, &score, (stamp && k < 64) ? &ap->stamps[8 * k + 4] : nullptr, tt);
      if (tt == 0) s_score = score;
      // (thread 0 wrote the score and run0_len, other threads the rest of s_gi before the body's last barrier: the four
      // publishing lanes are thread 0's wave, behind it in program order)
      if (tt < 4) gm_gran_store(mine4 + tt, tt, s_score, s_gi, tag);
      if (stamp && k < 64) ap->stamps[8 * k + 5] = wall_clock64();
    } else if (tt < 4) {
      // (nothing to score: the tags go out all the same -- the sweepers wait for EVERY workgroup of the grid in every
      // super-step, so nobody is ever more than one super-step behind: hc_resident.hip)
      gm_gran_store(mine4 + tt, tt, 0.0, GmPoseInfo{}, tag);
    }
    // ---- waves 0..G-1: the four granules of one slot per lane -- one 64-byte line --, all slots of the tree, into LDS.
    // (r04 had one granule KIND per wave: every line of the exchange block was asked for by four waves of every
    // workgroup in every poll, and "publish -> all here" of a lone chain was 3.8 us against the 1-cell form's 1.4.)
    if (wave < G) {
      const int n_grid = (int)gridDim.x - 1;  // (+ the bookkeeping workgroup)
      const int i = lane + 64 * wave;
      const int j = i < n_grid ? i : kHcSlots - 1;
      const HcGranule *const g0 = gran + ((size_t)pk * kRow + j) * 4;
      unsigned spins = 0;
      bool failed = false;
      for (;;) {
        u32x4 g[4];
        const HcGranule *gp[4] = {g0, g0 + 1, g0 + 2, g0 + 3};
        gran_fetch(g, gp);
        bool ok = true;
#pragma unroll
        for (int q = 0; q < 4; ++q) ok = ok && gran_tag(g[q]) == (tag & 0xffffu);
        if (ok) {
          GmPoseInfo gi;
          s_sc[j] = gran_score(g[0]);
          gi.run0_len = (int)g[0].z;
          gi.v0 = gran_score(g[1]);
          gi.last_head = (int)g[1].z;
          gi.last_v = gran_score(g[2]);
          gi.first_cx = (int)g[2].z;
          const unsigned long long w = (unsigned long long)__double_as_longlong(gran_score(g[3]));
          gi.first_cy = (int)(unsigned)w;
          gi.last_cx = (int)(unsigned)(w >> 32);
          gi.last_cy = (int)g[3].z;
          s_info[j] = gi;
        }
        if (__all(ok)) break;
        ++spins;
        if ((spins & 31u) == 0u) {
          const bool gone = __hip_atomic_load(&rc->fail_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ap->epoch;
          if (gone || spins > ap->spin_limit) {
            failed = true;
            break;
          }
        }
      }
      if (failed && lane == 0) s_sweep_failed = 1;
    }
    __syncthreads();  // (D) the four kinds of every slot are in LDS
    if (stamp && k < 64) ap->stamps[8 * k + 1] = wall_clock64();
    if (s_sweep_failed || k + 1 >= kHcResidentMaxSteps) {
      if (tt == 0) {
        __hip_atomic_store(&rc->fail_epoch, ap->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&host->error, s_sweep_failed ? 4 : 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __threadfence_system();
        __hip_atomic_store(&host->done_seq, ap->epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      break;  // (uniform: every thread read the same words behind the barrier)
    }
    if (wave == 0) {
      // ---- replay of super-step k's tree, lane = round instance (hc_chain.hip's GMapping replay)
      const HcState &sp = s_st;
      const int n_inst = (int)((ap->n_inst >> (8 * sp.shape)) & 0xffull);
      HcInst me;
      {
        const unsigned long long *src = &ap->shapes[sp.shape].inst[lane].w[0];
#pragma unroll
        for (int q = 0; q < 14; ++q) me.w[q] = src[q];
      }
      const bool active = lane < n_inst;
      const bool reach = active && (hc_is_root(me) || sp.failed + hc_nfail_parent(me) < ap->max_failed);
      const bool trailing = reach && hc_trailing(sp.failed + hc_nfail(me), ap->max_failed);
      double root_prob = sp.first ? s_sc[kHcSlots - 1] : sp.best_prob;
      HcCarry root_carry{sp.carry_cx, sp.carry_cy, sp.carry_prob};
      double first_raw = 0.0;
      if (sp.first) {
        first_raw = s_sc[kHcSlots - 1];
        root_prob = hc_gm_fix(s_sc[kHcSlots - 1], root_carry, s_info[kHcSlots - 1], scan);
      }
      double s6[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) s6[c] = s_sc[6 * lane + c];
      // the cache as this round's candidates meet it, in order -- corrected scores go back to LDS, where descendants
      // look up the score their path entered with
      HcCarry carry_after0 = root_carry, carry_after5 = root_carry;
      bool degenerate = false;
      if (active) {
        HcCarry cr = root_carry;
        const int par = hc_parent(me);
        if (par >= 0) {
          const GmPoseInfo &pg = s_info[6 * par + 5];
          degenerate = pg.last_head == 0;
          cr = HcCarry{pg.last_cx, pg.last_cy, pg.last_v};
        }
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          s6[c] = hc_gm_fix(s6[c], cr, s_info[6 * lane + c], scan);
          s_sc[6 * lane + c] = s6[c];
          if (c == 0) carry_after0 = cr;
        }
        carry_after5 = cr;
      }
      const int bpi = hc_bp_inst(me);
      const int bp_slot = (!active || bpi < 0) ? -1 : 6 * bpi + hc_bp_cand(me);
      const double enter = bp_slot < 0 ? root_prob : s_sc[bp_slot];
      double run = enter;
      int out = 0;
      unsigned accmask = 0u;
      // checked default mode (r06): a comparison whose two canonical sums lie within 2^-40 (relative) of each other is one
      // the tree sum with the device's exp cannot settle the way the reference's beam-order sum with glibc's exp would
      // -- reported (error 7), and the host redoes the match in the exact mode (exact_kernels.hip).  Two zero scores
      // are sums of zeros in any order: settled.
      bool unsettled = false;
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const bool live = c == 0 || !trailing;
        const double s = s6[c];
        if (ap->verify && live) {
          const double diff = __builtin_fabs(s - run);
          const double as = __builtin_fabs(s), ab = __builtin_fabs(run);
          unsettled = unsettled || (diff <= (as > ab ? as : ab) * 9.094947017729282e-13 && !(s == 0.0 && run == 0.0));
        }
        if (live && run < s) {  // strict: ties are rejections (pose_enumeration_scan_matcher.h:58)
          run = s;
          out = c + 1;
          accmask |= 1u << c;
        }
      }
      bool valid = reach;
#pragma unroll
      for (int o = 0; o < 7; ++o) {
        const unsigned long long has = __ballot(reach && out == o);
        valid = valid && (me.w[o] & ~has) == 0ull;
      }
      const bool terminal = valid && (trailing || hc_child(me, out) < 0);
      const unsigned long long tmask = __ballot(terminal);
      const int tl = tmask ? __ffsll((long long)tmask) - 1 : 0;
      if (stamp && k < 64) ap->stamps[8 * k + 7] = wall_clock64();
      // (the round states are made AFTER the decisions here, unlike hc_resident.hip: the cache fix-ups above are at
      // the register limit of a 1024-thread workgroup, and eleven live registers less is the difference to spilling)
      HcRound rr{sp.x, sp.y, sp.theta, sp.dt, sp.dr, sp.failed};
      if (valid) rr = hc_round_of(sp, me);
      HcRound rt;
      rt.x = bcast(rr.x, tl);
      rt.y = bcast(rr.y, tl);
      rt.theta = bcast(rr.theta, tl);
      rt.dt = bcast(rr.dt, tl);
      rt.dr = bcast(rr.dr, tl);
      rt.failed = (unsigned)bcast_i((int)rr.failed, tl);
      const int out_t = bcast_i(out, tl);
      const double run_t = bcast(run, tl);
      HcState next = sp;
      {
        // (the terminal lane's instance record in every lane: hc_next_core reads its depth and its path's moves)
        HcInst mt;
        mt.w[9] = (unsigned long long)bcast_ll((long long)me.w[9], tl);
        hc_advance(sp, mt, rt, out_t, run_t, ap->max_failed, 6ll * n_inst + (sp.first ? 1 : 0), &next);
      }
      {
        // the cache after the walk's last scorer call: the terminal round's last candidate
        const HcCarry fin = trailing ? carry_after0 : carry_after5;
        next.carry_cx = bcast_i(fin.cx, tl);
        next.carry_cy = bcast_i(fin.cy, tl);
        next.carry_prob = bcast(fin.prob, tl);
        // a pose whose whole scan is one run sat on the path: what it leaves in the cache depends on what it met --
        // the host-driven path redoes the match (error 3)
        if (__ballot(valid && degenerate) != 0ull && init_slot && lane == 0) host->error = 3;
        if (__ballot(valid && unsettled) != 0ull && init_slot && lane == 0) host->error = 7;
      }
      if (tmask == 0ull) {
        next.done = 1;
        if (init_slot && lane == 0) host->error = 1;
      }
      if (stamp && k < 64) ap->stamps[8 * k + 2] = wall_clock64();
      if (init_slot) {
        if (sp.first && lane == 0) {  // what the filter's cross-particle cache check looks at
          s_first_info = s_info[kHcSlots - 1];
          s_first_raw = first_raw;
        }
        if (ap->trace) {
          HcTraceEntry *const trace = ap->trace + (size_t)blockIdx.y * (size_t)ap->trace_stride;
          const long long base = sp.calls + (sp.first ? 1 : 0);
          if (sp.first && lane == 0 && ap->trace_cap > 0) {
            HcTraceEntry e{sp.x, sp.y, sp.theta, root_prob, 1, 0};
            trace[0] = e;
          }
          if (valid) {
            const int nc = trailing ? 1 : 6;
            for (int c = 0; c < nc; ++c) {
              HcTraceEntry e;
              hc_candidate(rr.x, rr.y, rr.theta, rr.dt, rr.dr, c, &e.x, &e.y, &e.theta);
              e.score = s_sc[6 * lane + c];
              e.accepted = (accmask >> c) & 1u;
              e.pad = 0;
              const long long at = base + 6ll * hc_depth(me) + c;
              if (at < ap->trace_cap) trace[at] = e;
              else host->error = 2;
            }
          }
        }
        if (next.done) {
          __threadfence_system();
          if (lane == 0) {
            HcHostOut *h = host;
            h->pose[0] = next.x;
            h->pose[1] = next.y;
            h->pose[2] = next.theta;
            h->best_prob = next.best_prob;
            h->calls = next.calls;
            h->evaluated = next.evaluated;
            h->steps = next.steps;
            h->rescored = 0;
            h->gm_cx = next.carry_cx;
            h->gm_cy = next.carry_cy;
            h->gm_prob = next.carry_prob;
            h->first_info = s_first_info;
            h->first_raw = s_first_raw;
            __hip_atomic_store(&h->done_seq, ap->epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            if (ap->n_done) {
              __threadfence_system();
              const unsigned before = atomicAdd(ap->n_done, 1u);
              if (before + 1u == gridDim.y && ap->h_all_done)
                __hip_atomic_store(ap->h_all_done, ap->epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
          }
        }
      }
      if (lane == 0) {
        // (field by field, only what a super-step changes: the struct assigned as a whole went through scratch -- see
        // hc_resident.hip)
        HcState &w = s_st;
        w.x = next.x;
        w.y = next.y;
        w.theta = next.theta;
        w.best_prob = next.best_prob;
        w.dt = next.dt;
        w.dr = next.dr;
        w.recent_acc = next.recent_acc;
        w.recent_n = next.recent_n;
        w.calls = next.calls;
        w.evaluated = next.evaluated;
        w.failed = next.failed;
        w.shape = next.shape;
        w.done = next.done;
        w.first = next.first;
        w.steps = next.steps;
        w.mode = next.mode;
        w.carry_cx = next.carry_cx;
        w.carry_cy = next.carry_cy;
        w.carry_prob = next.carry_prob;
      }
    }
  }
}

#define HCRG_LAUNCH(NTV)                                                                                        \
  do {                                                                                                          \
    if (grid <= 64) {                                                                                           \
      if (e0 || e1)                                                                                             \
        hipExtLaunchKernelGGL((k_hc_chain_resident_gm<NTV, KB, 1>), dim3(grid, n_chains), dim3(NTV), shm, stream, e0, e1, 0, a); \
      else                                                                                                      \
        hipLaunchKernelGGL((k_hc_chain_resident_gm<NTV, KB, 1>), dim3(grid, n_chains), dim3(NTV), shm, stream, a);   \
    } else {                                                                                                    \
      if (e0 || e1)                                                                                             \
        hipExtLaunchKernelGGL((k_hc_chain_resident_gm<NTV, KB, 4>), dim3(grid, n_chains), dim3(NTV), shm, stream, e0, e1, 0, a); \
      else                                                                                                      \
        hipLaunchKernelGGL((k_hc_chain_resident_gm<NTV, KB, 4>), dim3(grid, n_chains), dim3(NTV), shm, stream, a);   \
    }                                                                                                           \
  } while (0)

template <int KB>
static hipError_t launch_res_gm(const HcChainArgs &a, int nt, hipStream_t stream, hipEvent_t e0, hipEvent_t e1,
                                int n_chains) {
  const int grid = 6 * a.max_inst + 1;
  if (grid > 256) return hipErrorInvalidValue;
  // (the dynamic LDS of hc_chain.hip's launch_gm: K3's arrays + one double per thread for the helper lanes)
  const size_t shm = (size_t)KB * 256 * sizeof(double) + 4 * KB * sizeof(int2) + 4 * KB * sizeof(int) +
                     2 * (size_t)KB * 256 * sizeof(int) + (nt >= 512 ? (size_t)nt * sizeof(double) : 0);
  if (nt == 1024) HCRG_LAUNCH(1024);
  else if (nt == 256) HCRG_LAUNCH(256);
  else HCRG_LAUNCH(512);
  return hipGetLastError();
}
#undef HCRG_LAUNCH

hipError_t launch_hc_chain_resident_gm(const HcChainArgs &a, int nt, hipStream_t stream, hipEvent_t e0, hipEvent_t e1,
                                       int n_chains) {
  if (!a.rctl_gm || a.seq || a.jobs) return hipErrorInvalidValue;
  switch ((a.scan.n + 255) / 256) {
    case 1: return launch_res_gm<1>(a, nt, stream, e0, e1, n_chains);
    case 2: return launch_res_gm<2>(a, nt, stream, e0, e1, n_chains);
    case 3: return launch_res_gm<3>(a, nt, stream, e0, e1, n_chains);
    case 4: return launch_res_gm<4>(a, nt, stream, e0, e1, n_chains);
    case 5: return launch_res_gm<5>(a, nt, stream, e0, e1, n_chains);
    default: break;
  }
  return hipErrorInvalidValue;
}

// workgroups of `nt` threads the device keeps resident at once (see hc_resident_capacity)
hipError_t hc_resident_gm_capacity(int nt, int n_beams, int *out_wgs, int *out_per_cu) {
  int dev = 0, cus = 0, per_cu = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (e != hipSuccess) return e;
  const int kb = (n_beams + 255) / 256;
  const size_t shm = (size_t)kb * 256 * sizeof(double) + 4 * kb * sizeof(int2) + 4 * kb * sizeof(int) +
                     2 * (size_t)kb * 256 * sizeof(int) + (nt >= 512 ? (size_t)nt * sizeof(double) : 0);
  const void *fn = nt == 1024 ? (const void *)k_hc_chain_resident_gm<1024, 5, 4>
                              : (nt == 256 ? (const void *)k_hc_chain_resident_gm<256, 5, 4>
                                           : (const void *)k_hc_chain_resident_gm<512, 5, 4>);
  e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, nt, shm);
  if (e != hipSuccess) return e;
  const int by_waves = nt == 256 ? 6 : 2048 / nt;  // (three 168-VGPR waves per SIMD at 256 threads, four 128-VGPR ones else)
  per_cu = per_cu < by_waves ? per_cu : by_waves;
  if (nt == 256 && per_cu > 3) per_cu = 3;
  if (per_cu > 6) per_cu = 6;
  *out_wgs = per_cu * (cus - 1);  // (one CU's worth of margin: hc_resident_capacity)
  if (out_per_cu) *out_per_cu = per_cu;
  return hipSuccess;
}

}  // namespace slamhip
