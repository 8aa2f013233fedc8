// hc_chain.hip -- one hill-climbing process_scan as a chain of kernels with no host in between
// (design and the functions shared with the CPU test: hc_chain.h).
//
// Kernel k_hc_chain_step, launched K times back to back on the context's stream with k = 0, 1, 2, ...:
//   prologue (k > 0)  replay of super-step k-1: every workgroup's wave 0 walks the scores of the previous
//                     tree (one lane per round instance) and derives the root state of this super-step;
//                     the last workgroup also stores it (for kernel k+1), writes the trace entries of the
//                     walked rounds and, when the enumerator has run out, publishes the result to the host
//   body              workgroup w scores ONE pose: candidate w % 6 of round instance w / 6 of this
//                     super-step's tree (the last workgroup: the initial pose, first super-step only) with
//                     the arithmetic of k_score_point (score_device.h) and the canonical 256-partial sum,
//                     so its bits equal the host-driven matcher's
// Kernels launched past the end of the chain see `done_epoch == epoch` and return.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "gm_score_device.h"
#include "hc_chain_device.h"
#include "score_device.h"

namespace slamhip {

static constexpr int kSumLanes = 256;  // the canonical sum's partials (= k_score_point's block)

// value of lane `lane` (uniform) in every lane: v_readlane, no trip through the LDS crossbar
__device__ __forceinline__ int bcast_i(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ long long bcast_ll(long long v, int lane) {
  const int lo = bcast_i((int)(unsigned)(unsigned long long)v, lane);
  const int hi = bcast_i((int)(unsigned)((unsigned long long)v >> 32), lane);
  return (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo);
}
__device__ __forceinline__ double bcast(double v, int lane) {
  return __longlong_as_double(bcast_ll(__double_as_longlong(v), lane));
}
// Kernel arguments are fetched where they are first used: a scalar load in the middle of the replay that
// misses the scalar cache stalls its wave for hundreds of nanoseconds (three of them made half of the
// replay's 1.9 us).  PIN makes the value exist in a scalar register HERE, i.e. next to the staging loads
// at kernel entry, whose latency hides it.
#define HC_PIN32(x) asm volatile("" ::"s"(x))
#define HC_PIN64(x) asm volatile("" ::"s"((unsigned long long)(x)))
#define HC_PINF(x) asm volatile("" ::"s"(__double_as_longlong(x)))

// LDS copy of the previous tree's round instances: 17-word (136-byte) stride -- 128 would put the same
// word of every lane's record in one bank
static constexpr int kWalkStride = 17;

// GMapping OOPE in the chain: what the cross-pose cache (Q19) does to a replayed pose's score, and what the
// pose leaves in it -- MatchJob's gm_apply_carry (matchers.h) for one pose
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

// MODEL: SLAMHIP_CELL_OCC / _TBM = the 1-cell OOPE (k_score_point's arithmetic), SLAMHIP_CELL_GMAPPING = the
// GMapping OOPE (K3's one-pose body, KB = ceil(beams / 256))
// BATCH: grid.y independent matches, each with its own map and scan (HcChainArgs::jobs)
template <int MODEL, int NT, bool SEQ, int KB, bool BATCH>
__global__ __launch_bounds__(NT) void k_hc_chain_step(HcChainArgs a, int k) {
  constexpr bool GM = MODEL == SLAMHIP_CELL_GMAPPING;
  static_assert(!(BATCH && GM), "the filter's chains share one map view and one scan");
  extern __shared__ double s_term[];            // point OOPE: one term per beam; GMapping: K3's arrays
  __shared__ GmPoseInfo s_info[GM ? kHcSlots : 1];  // side outputs of the previous tree's poses
  __shared__ unsigned long long s_hash[GM ? 1 : kHcSlots + 7];  // term-vector fingerprints of the previous tree's poses
  __shared__ unsigned long long s_hpart[4];
  __shared__ double s_unknown[4];
  __shared__ int s_run0_len;
  __shared__ double s_sc[kHcSlots + 7];         // scores of the previous tree
  __shared__ unsigned long long s_walk[kHcMaxInst * kWalkStride];
  __shared__ HcState s_prev;                    // root state of the previous super-step
  __shared__ HcInst s_mine[kHcShapes];          // this workgroup's round instance in every shape
  __shared__ double s_pose[4];                  // x, y, sin, cos of the pose this workgroup scores
  __shared__ int s_go, s_mode;
  __shared__ double s_part[4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // the grid holds 6 x (instances of the largest shape) scoring workgroups and, last, the bookkeeping one, which
  // keeps slot kHcSlots - 1 whatever the grid size
  const int slot = blockIdx.x + 1 == gridDim.x ? kHcSlots - 1 : (int)blockIdx.x;
  HcChainCtl *ctl = a.ctl + blockIdx.y;  // (one chain per grid row)
  HcHostOut *host = a.host + blockIdx.y;
  // this chain's map and scan: kernel arguments, or -- a batch of matches -- its entry of the job table (a uniform
  // address read before any store of the kernel: scalar loads, in flight with the staging loads below)
  MapView map;
  ScanView scan;
  if (BATCH) {
    const HcJobView *__restrict__ jv = a.jobs + blockIdx.y;
    map = jv->map;
    scan = jv->scan;
  } else {
    map = a.map;
    scan = a.scan;
  }
  // ---- loads that depend on nothing: issued first, they overlap the replay below (the done test waits
  // for its word only after everything else is in flight)
  const bool stamp = SLAMHIP_STAMPS_ON(a.stamps && slot == 1 && t == 0 && k < 64);
  if (stamp) a.stamps[8 * k + 0] = wall_clock64();
  const unsigned done_epoch = __hip_atomic_load(&ctl->done_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int n = scan.n;
  const int inst_of_slot = slot / 6, cand = slot - 6 * inst_of_slot;
  const bool init_slot = slot == kHcSlots - 1;
  double br = 0.0, bc = 0.0, bs = 0.0, bw = 0.0, bf = 0.0;
  if (t < n) {
    br = scan.range[t];
    bc = scan.cos_a[t];
    bs = scan.sin_a[t];
    bw = scan.weight[t];
    bf = scan.factor[t];
  }
  if (wave == 1 && lane < kHcShapes && !init_slot) {
    const uint4 *src = reinterpret_cast<const uint4 *>(&a.shapes[lane].inst[inst_of_slot]);
    uint4 *dst = reinterpret_cast<uint4 *>(&s_mine[lane]);
#pragma unroll
    for (int q = 0; q < (int)(sizeof(HcInst) / 16); ++q) dst[q] = src[q];
  }
  HC_PIN32(a.max_failed);
  HC_PIN64(a.n_inst);
  HC_PIN64(a.host);
  HC_PIN64(a.inits);
  HC_PIN64(a.trace);
  HC_PIN32(a.trace_cap);
  HC_PIN64(a.stamps);
  HC_PIN64(map.payload);
  HC_PIN32(map.width);
  HC_PIN32(map.height);
  HC_PIN32(map.pitch);
  HC_PIN32(map.origin_x);
  HC_PIN32(map.origin_y);
  HC_PINF(map.scale);
  HC_PINF(map.inv_scale);
  HC_PINF(map.unknown[0]);
  HC_PINF(map.unknown[1]);
  HC_PINF(map.unknown[2]);
  HC_PINF(map.unknown[3]);
  HC_PIN32(a.oie);
  HC_PINF(scan.tot_w);
  const int pb = (k - 1) & 1;
  if (k > 0) {
    // what the previous tree left behind: the slots of the largest shape's instances, and the bookkeeping slot
    const int n_stage = 6 * a.max_inst;
    const double *sc_prev = ctl->scores[pb];
    for (int i = t; i <= n_stage; i += NT) {
      const int j = i < n_stage ? i : kHcSlots - 1;
      s_sc[j] = sc_prev[j];
    }
    const unsigned long long *wsrc = &ctl->walk[pb][0].w[0];
    for (int q = t; q < a.max_inst * 16; q += NT) s_walk[(q >> 4) * kWalkStride + (q & 15)] = wsrc[q];
    if (t < (int)(sizeof(HcState) / 8))
      reinterpret_cast<double *>(&s_prev)[t] = reinterpret_cast<const double *>(&ctl->state[pb])[t];
    if (!GM && a.verify) {
      for (int i = t; i <= n_stage; i += NT) {
        const int j = i < n_stage ? i : kHcSlots - 1;
        s_hash[j] = ctl->hashes[pb][j];
      }
    }
    if (GM) {
      constexpr int kInfoWords = (int)(sizeof(GmPoseInfo) / 8);
      const double *isrc = reinterpret_cast<const double *>(&ctl->infos[pb][0]);
      double *idst = reinterpret_cast<double *>(&s_info[0]);
      for (int q = t; q < (n_stage + 1) * kInfoWords; q += NT) {
        const int j = q < n_stage * kInfoWords ? q : q - n_stage * kInfoWords + (kHcSlots - 1) * kInfoWords;
        idst[j] = isrc[j];
      }
    }
  }
  if (GM && t == 64) {
    s_unknown[0] = map.unknown[0];
    s_unknown[1] = map.unknown[1];
    s_unknown[2] = map.unknown[2];
  }
  if (GM && t == 65) s_run0_len = scan.n;
  if (done_epoch == a.epoch) return;  // launched past the end of the chain (uniform: before any barrier)
  __syncthreads();
  if (stamp) a.stamps[8 * k + 1] = wall_clock64();

  if (wave == 0) {
    HcState st;
    if (k == 0) {
      st = HcState{};
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
      if (init_slot) {
        if (lane == 0) {
          ctl->state[0] = st;
          __hip_atomic_store(&host->progress, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        ctl->walk[0][lane] = a.shapes[a.shape0].inst[lane];
      }
    } else {
      // ---- replay of the previous tree, lane = round instance
      const HcState sp = s_prev;
      HcInst me;  // the whole record at once: sixteen LDS loads in flight, fields by shifts
#pragma unroll
      for (int q = 0; q < 16; ++q) me.w[q] = s_walk[lane * kWalkStride + q];
      if (stamp) a.stamps[8 * k + 6] = wall_clock64();
      const int n_inst = (int)((a.n_inst >> (8 * sp.shape)) & 0xffull);
      double root_prob = sp.first ? s_sc[kHcSlots - 1] : sp.best_prob;
      HcCarry root_carry{sp.carry_cx, sp.carry_cy, sp.carry_prob};
      if (GM && sp.first) {
        if (init_slot && lane == 0) {  // what the filter's cross-particle cache check looks at
          ctl->first_info = s_info[kHcSlots - 1];
          ctl->first_raw = s_sc[kHcSlots - 1];
        }
        root_prob = hc_gm_fix(s_sc[kHcSlots - 1], root_carry, s_info[kHcSlots - 1], scan);
      }
      const bool active = lane < n_inst;
      const bool reach = active && (hc_is_root(me) || sp.failed + hc_nfail_parent(me) < a.max_failed);
      const bool trailing = reach && hc_trailing(sp.failed + hc_nfail(me), a.max_failed);
      double s6[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) s6[c] = s_sc[6 * lane + c];
      // GMapping: the cache as this round's candidates meet it, in order -- corrected scores go back to LDS,
      // where descendants look up the score their path entered with
      HcCarry carry_after0 = root_carry, carry_after5 = root_carry;
      bool degenerate = false;
      if (GM && active) {
        HcCarry cr = root_carry;
        const int par = hc_parent(me);
        if (par >= 0) {
          // the cache after the parent round's last candidate.  What a pose leaves behind depends on what it
          // met only when its whole scan is one run (last_head == 0): such a parent is handed to the host
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
      const double enter = bp_slot < 0 ? root_prob : s_sc[bp_slot];  // canonical (reported) score entering the round
      double run = enter;
      int out = 0;
      unsigned accmask = 0u;
      bool ambiguous = false;
      unsigned long long run_hash = 0ull;
      const bool rescored = !GM && a.verify && sp.mode == 1;  // decisions from the beam-order sums of this tree
      if (!GM && a.verify) {
        // slot kHcSlots-1 holds the base pose of a re-scored tree (or the initial pose): its sums head the path
        const bool base_here = sp.first || sp.mode == 1;
        const unsigned long long root_hash = base_here ? s_hash[kHcSlots - 1] : sp.best_hash;
        unsigned long long hb = bp_slot < 0 ? root_hash : s_hash[bp_slot];
        unsigned long long h6[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) h6[c] = s_hash[6 * lane + c];
        if (!rescored) {
          // decisions from the canonical sums; a comparison closer than the two summation orders can differ
          // (2^-40, relative) between poses whose term vectors differ is one the tree sum cannot settle
#pragma unroll
          for (int c = 0; c < 6; ++c) {
            const double s = s6[c];
            const double diff = __builtin_fabs(s - run);
            const double as = __builtin_fabs(s), ab = __builtin_fabs(run);
            const bool live = c == 0 || !trailing;
            const bool close = diff <= (as > ab ? as : ab) * 9.094947017729282e-13;  // NaN: false, a rejection
            // (equal fingerprints with different sums: not identical vectors -- a collision, equally unsettled)
            ambiguous = ambiguous || (live && close && (h6[c] != hb || __double_as_longlong(s) != __double_as_longlong(run)));
            const bool acc = live && run < s;  // strict: ties are rejections (pose_enumeration_scan_matcher.h:58)
            run = acc ? s : run;
            hb = acc ? h6[c] : hb;
            out = acc ? c + 1 : out;
            accmask |= acc ? 1u << c : 0u;
          }
        } else {
          // re-scored tree: the same comparisons on the beam-order sums (read where they lie: this is the rare step)
          const double *seq = ctl->scores_seq[pb];
          const double root_dec = seq[kHcSlots - 1];
          double bdec = bp_slot < 0 ? root_dec : seq[bp_slot];
#pragma unroll
          for (int c = 0; c < 6; ++c) {
            const double sd = active ? seq[6 * lane + c] : 0.0;
            const bool acc = (c == 0 || !trailing) && bdec < sd;
            bdec = acc ? sd : bdec;
            run = acc ? s6[c] : run;  // the canonical sum of the pose accepted last: stored and reported
            hb = acc ? h6[c] : hb;
            out = acc ? c + 1 : out;
            accmask |= acc ? 1u << c : 0u;
          }
        }
        run_hash = hb;
      } else {
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          const bool live = c == 0 || !trailing;
          if (GM && a.verify && live) {
            // checked default mode over the GMapping OOPE (r06): see hc_resident_gm.hip -- a comparison within 2^-40 is
            // reported (error 7) and the host redoes the match in the exact mode
            const double s = s6[c];
            const double diff = __builtin_fabs(s - run);
            const double as = __builtin_fabs(s), ab = __builtin_fabs(run);
            ambiguous = ambiguous || (diff <= (as > ab ? as : ab) * 9.094947017729282e-13 && !(s == 0.0 && run == 0.0));
          }
          if (live && run < s6[c]) {  // strict: ties are rejections
            run = s6[c];
            out = c + 1;
            accmask |= 1u << c;
          }
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
      // exactly one lane is terminal: the walk's last round
      const int tl = tmask ? __ffsll((long long)tmask) - 1 : 0;
      if (stamp) a.stamps[8 * k + 7] = wall_clock64();
      // checked default mode: a comparison on the walked path that the tree sum cannot settle -> the same tree is
      // scored once more, in beam order as well, and decided from those sums
      const bool dirty = !GM && a.verify && sp.mode == 0 && __ballot(valid && ambiguous) != 0ull;
      HcState next = sp;
      if (terminal && !dirty) {
        const HcRound r = hc_round_of(sp, me);
        hc_advance(sp, me, r, out, run, a.max_failed, 6ll * n_inst + (sp.first ? 1 : 0), &next);
      }
      // hand the new root state to every lane
      next.x = bcast(next.x, tl);
      next.y = bcast(next.y, tl);
      next.theta = bcast(next.theta, tl);
      next.best_prob = bcast(next.best_prob, tl);
      next.dt = bcast(next.dt, tl);
      next.dr = bcast(next.dr, tl);
      next.recent_acc = bcast(next.recent_acc, tl);
      next.recent_n = bcast(next.recent_n, tl);
      next.calls = bcast_ll(next.calls, tl);
      next.evaluated = bcast_ll(next.evaluated, tl);
      next.failed = (unsigned)bcast_i((int)next.failed, tl);
      next.shape = bcast_i(next.shape, tl);
      next.done = bcast_i(next.done, tl);
      next.first = 0;
      next.steps = sp.steps + 1;
      if (!GM && a.verify) {
        next.best_hash = (unsigned long long)bcast_ll((long long)run_hash, tl);
        next.mode = 0;
        next.rescored = sp.rescored;
      }
      if (dirty) {
        next = sp;  // same root, same tree, same `first`
        next.mode = 1;
        next.steps = sp.steps + 1;
        next.evaluated = sp.evaluated + 6ll * n_inst + 1;
        next.rescored = sp.rescored + 1;
      }
      if (GM) {
        // the cache after the walk's last scorer call: the terminal round's last candidate
        const HcCarry fin = trailing ? carry_after0 : carry_after5;
        next.carry_cx = bcast_i(fin.cx, tl);
        next.carry_cy = bcast_i(fin.cy, tl);
        next.carry_prob = bcast(fin.prob, tl);
        if (__ballot(valid && degenerate) != 0ull && init_slot && lane == 0) host->error = 3;
        if (__ballot(valid && ambiguous) != 0ull && init_slot && lane == 0) host->error = 7;
      }
      if (tmask == 0ull) {  // cannot happen (the root round is always on the path): stop instead of looping
        next.done = 1;
        if (init_slot && lane == 0) host->error = 1;
      }
      // r06: the inert tail (hc_inert, hc_chain.h) -- a next root whose six candidates ARE that root, bit for bit: what the
      // reference still does is score that pose 6 x (limit - failed) + 1 more times, a tie and a rejection each.  The
      // chain ends here; the bookkeeping workgroup writes those scorer calls into the trace (below).  (The co-resident
      // form also ends on CERTIFIED roots, long before: hc_resident.hip.)
      long long tail_calls = 0;
      if (!GM && a.inert_tail && !dirty && !next.done && hc_inert(next.x, next.y, next.theta, next.dt, next.dr)) {
        tail_calls = 6ll * (long long)(a.max_failed - next.failed) + 1ll;
        next.done = 1;
      }
      st = next;
      if (init_slot) {
        // ---- the last workgroup keeps the books (it scores nothing after the first super-step, so the
        // dependent loads and stores below are on no pose's critical path)
        if (a.trace && !dirty) {
          HcTraceEntry *const trace = a.trace + (size_t)blockIdx.y * (size_t)a.trace_stride;
          const long long base = sp.calls + (sp.first ? 1 : 0);
          if (sp.first && lane == 0 && a.trace_cap > 0) {
            HcTraceEntry e{sp.x, sp.y, sp.theta, root_prob, 1, 0};
            trace[0] = e;
          }
          if (valid) {
            const HcRound r = hc_round_of(sp, me);
            const int nc = trailing ? 1 : 6;
            for (int c = 0; c < nc; ++c) {
              HcTraceEntry e;
              hc_candidate(r.x, r.y, r.theta, r.dt, r.dr, c, &e.x, &e.y, &e.theta);
              e.score = s_sc[6 * lane + c];
              e.accepted = (accmask >> c) & 1u;
              e.pad = 0;
              const long long at = base + 6ll * hc_depth(me) + c;
              if (at < a.trace_cap) trace[at] = e;
              else host->error = 2;
            }
          }
        }
        if (tail_calls > 0 && a.trace) {
          HcTraceEntry *const trace = a.trace + (size_t)blockIdx.y * (size_t)a.trace_stride;
          for (long long q = lane; q < tail_calls; q += 64) {
            // scorer call q of the tail: candidate q % 6 of its round q / 6, whose steps were halved that often
            const double hlf = hc_pow_half((unsigned)(q / 6));
            HcTraceEntry e;
            hc_candidate(next.x, next.y, next.theta, next.dt * hlf, next.dr * hlf, (int)(q % 6), &e.x, &e.y, &e.theta);
            e.score = next.best_prob;
            e.accepted = 0;
            e.pad = 0;
            if (next.calls + q < a.trace_cap) trace[next.calls + q] = e;
            else host->error = 2;
          }
        }
        next.calls += tail_calls;
        if (lane == 0) ctl->state[k & 1] = next;
        if (!next.done) ctl->walk[k & 1][lane] = a.shapes[next.shape].inst[lane];
        if (next.done) {
          // every lane's trace stores first, then the result, then the flag the host spins on
          __threadfence_system();
          if (lane == 0) {
            ctl->done_epoch = a.epoch;
            HcHostOut *h = host;
            h->pose[0] = next.x;
            h->pose[1] = next.y;
            h->pose[2] = next.theta;
            h->best_prob = next.best_prob;
            h->calls = next.calls;
            h->tail_calls = tail_calls;
            h->evaluated = next.evaluated;
            h->steps = next.steps;
            h->rescored = next.rescored;
            h->gm_cx = next.carry_cx;
            h->gm_cy = next.carry_cy;
            h->gm_prob = next.carry_prob;
            if (GM) {
              h->first_info = ctl->first_info;
              h->first_raw = ctl->first_raw;
            }
            if (a.n_done) atomicAdd(a.n_done, 1u);
            __hip_atomic_store(&h->done_seq, a.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
          }
        } else if (lane == 0) {
          __hip_atomic_store(&host->progress, (unsigned)(k + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
    if (stamp) a.stamps[8 * k + 2] = wall_clock64();
    // ---- this workgroup's pose
    bool go = !st.done;
    double px = st.x, py = st.y, pth = st.theta;
    if (init_slot) {
      go = go && (st.first || st.mode == 1);  // the initial pose / the base pose of a re-scored tree
    } else if (go) {
      HcInst in;
#pragma unroll
      for (int q = 0; q < 14; ++q) in.w[q] = s_mine[st.shape].w[q];
      go = inst_of_slot < (int)((a.n_inst >> (8 * st.shape)) & 0xffull) &&
           (hc_is_root(in) || st.failed + hc_nfail_parent(in) < a.max_failed);  // else: behind the end of the chain
      if (go) {
        const HcRound r = hc_round_of(st, in);
        go = !(hc_trailing(r.failed, a.max_failed) && cand > 0);  // a trailing round has one candidate
        hc_candidate(r.x, r.y, r.theta, r.dt, r.dr, cand, &px, &py, &pth);
      }
    }
    if (go) {
      double sn, cs;
      sincos(pth, &sn, &cs);
      if (lane == 0) {
        s_pose[0] = px;
        s_pose[1] = py;
        s_pose[2] = sn;
        s_pose[3] = cs;
      }
    }
    if (lane == 0) {
      s_go = go ? 1 : 0;
      s_mode = st.mode;
    }
  }
  __syncthreads();
  if (stamp) a.stamps[8 * k + 3] = wall_clock64();
  if (!s_go) return;
  const double px = s_pose[0], py = s_pose[1], sn = s_pose[2], cs = s_pose[3];

  if (GM) {
    // ---- the GMapping OOPE: K3's one-pose body (gm_score_device.h); the side outputs of the cross-pose cache
    // go next to the score, the replay of the next kernel applies the cache in the reference's call order
    constexpr int KBG = KB > 0 ? KB : 1;
    double score = 0.0;
    const int *tiles = a.tables ? a.tables + (size_t)a.slots[blockIdx.y] * a.table_stride : nullptr;
    gm_score_pose_wide<KBG, NT>(map, scan, a.gm, tiles, s_unknown, px, py, sn, cs, br, bc, bs, s_term, &s_run0_len,
                                s_part, &ctl->infos[k & 1][slot], &score, stamp ? &a.stamps[8 * k + 4] : nullptr);
    if (t == 0) {
      ctl->scores[k & 1][slot] = score;
      if (stamp) a.stamps[8 * k + 5] = wall_clock64();
    }
    return;
  }
  // ---- score it: terms by beam, then the canonical sum (256 strided partials in ascending beam order,
  // wave butterfly, (g0+g1)+(g2+g3)) -- k_score_point's order
  // up to four beams per thread at a time: their cell gathers are issued together (a clamped index keeps the
  // code free of branches), the probabilities follow when the values arrive
  for (int base = t; base < n; base += 4 * NT) {
    double4 cell[4];
    double w_[4], f_[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int b = base + j * NT;
      w_[j] = 0.0;
      f_[j] = 0.0;
      cell[j] = make_double4(0.0, 0.0, 0.0, 0.0);
      if ((base - lane) + j * NT >= n) continue;  // no lane of this wave has a beam in this slot
      const int bc_ = b < n ? b : n - 1;
      double r_ = br, ca = bc, sa = bs;
      w_[j] = bw;
      f_[j] = bf;
      if (j > 0 || base != t) {
        r_ = scan.range[bc_];
        ca = scan.cos_a[bc_];
        sa = scan.sin_a[bc_];
        w_[j] = scan.weight[bc_];
        f_[j] = scan.factor[bc_];
      }
      cell[j] = beam_cell<MODEL>(map, px, py, sn, cs, r_, ca, sa);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int b = base + j * NT;
      if (b < n) s_term[b] = cell_probability<MODEL>(a.oie, cell[j]) * w_[j] * f_[j];
    }
  }
  __syncthreads();
  if (stamp) a.stamps[8 * k + 4] = wall_clock64();
  if (SEQ) {
    // the reference's own order: one running sum over the beams (weighted_mean_point_probability_spe.h:108-124)
    if (t == 0) {
      double acc = 0.0;
      int b = 0;
      for (; b + 8 <= n; b += 8) {
        const double t0 = s_term[b], t1 = s_term[b + 1], t2 = s_term[b + 2], t3 = s_term[b + 3];
        const double t4 = s_term[b + 4], t5 = s_term[b + 5], t6 = s_term[b + 6], t7 = s_term[b + 7];
        acc = acc + t0;
        acc = acc + t1;
        acc = acc + t2;
        acc = acc + t3;
        acc = acc + t4;
        acc = acc + t5;
        acc = acc + t6;
        acc = acc + t7;
      }
      for (; b < n; ++b) acc = acc + s_term[b];
      ctl->scores[k & 1][slot] = (scan.tot_w == 0.0) ? __builtin_nan("") : acc / scan.tot_w;
    }
    return;
  }
  const bool verify = !GM && a.verify;
  if (t < kSumLanes) {
    double acc = 0.0;
    // fingerprint of the term vector (score_device.h)
    unsigned long long h = 0ull;
    unsigned k_lo = (2u * (unsigned)t + 1u) * 0x9E3779B1u, k_hi = (2u * (unsigned)t + 1u) * 0x85EBCA6Bu;
    for (int b = t; b < n; b += kSumLanes) {
      const double term = s_term[b];
      acc = acc + term;
      if (verify) {
        h += term_fingerprint(term, k_lo, k_hi);
        k_lo += 2u * kSumLanes * 0x9E3779B1u;
        k_hi += 2u * kSumLanes * 0x85EBCA6Bu;
      }
    }
    wave_xor_sum_with(acc, h);
    if (lane == 0) {
      s_part[wave] = acc;
      s_hpart[wave] = h;
    }
  }
  __syncthreads();
  if (t == 0) {
    const double total = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
    ctl->scores[k & 1][slot] = (scan.tot_w == 0.0) ? __builtin_nan("") : total / scan.tot_w;
    if (verify) {
      ctl->hashes[k & 1][slot] = s_hpart[0] + s_hpart[1] + s_hpart[2] + s_hpart[3];
    }
    if (stamp) a.stamps[8 * k + 5] = wall_clock64();
  }
  if (verify && s_mode && t == 64) {
    // re-scored super-step: the reference's own order as well, one running sum over the beams
    double acc = 0.0;
    for (int b = 0; b < n; ++b) acc = acc + s_term[b];
    ctl->scores_seq[k & 1][slot] = (scan.tot_w == 0.0) ? __builtin_nan("") : acc / scan.tot_w;
  }
}

// e0 / e1 (optional): HIP events attached to the dispatch (kernel begin .. end), see SLAMHIP_LAUNCH in
// score_kernels.hip
#define HC_LAUNCH(NTV)                                                                                          \
  do {                                                                                                          \
    if (e0 || e1)                                                                                               \
      hipExtLaunchKernelGGL((k_hc_chain_step<MODEL, NTV, SEQ, KB, BATCH>), dim3(grid, n_chains), dim3(NTV), shm, stream, e0, e1, 0, a, k); \
    else                                                                                                        \
      hipLaunchKernelGGL((k_hc_chain_step<MODEL, NTV, SEQ, KB, BATCH>), dim3(grid, n_chains), dim3(NTV), shm, stream, a, k);     \
  } while (0)

template <int MODEL, bool SEQ, bool BATCH>
static hipError_t launch_nt(const HcChainArgs &a, int k, int nt, hipStream_t stream, hipEvent_t e0, hipEvent_t e1,
                            int n_chains) {
  constexpr int KB = 0;
  const int grid = 6 * a.max_inst + 1;
  const size_t shm = sizeof(double) * (size_t)(a.scan.n > 0 ? a.scan.n : 1);
  switch (nt) {
    case 256: HC_LAUNCH(256); break;
    case 1024: HC_LAUNCH(1024); break;
    default: HC_LAUNCH(512); break;
  }
  return hipGetLastError();
}

// GMapping OOPE: 512 threads per pose (two workgroups per CU: all slots of a 64-instance super-step resident at
// once) or 1024 (one per CU: trees of at most 42 instances)
template <int KB>
static hipError_t launch_gm(const HcChainArgs &a, int k, int nt, hipStream_t stream, hipEvent_t e0, hipEvent_t e1,
                            int n_chains) {
  constexpr int MODEL = SLAMHIP_CELL_GMAPPING;
  constexpr bool SEQ = false, BATCH = false;
  const int grid = 6 * a.max_inst + 1;
  // (+ one double per thread for the helper lanes of gm_score_pose_wide: its 512- and 1024-thread forms)
  const size_t shm = (size_t)KB * 256 * sizeof(double) + 4 * KB * sizeof(int2) + 4 * KB * sizeof(int) +
                     2 * (size_t)KB * 256 * sizeof(int) + (nt >= 512 ? (size_t)nt * sizeof(double) : 0);
  if (nt == 1024) HC_LAUNCH(1024);
  else if (nt == 256) HC_LAUNCH(256);
  else HC_LAUNCH(512);
  return hipGetLastError();
}
#undef HC_LAUNCH

__global__ void k_chain_marker(const unsigned *n_done, unsigned *h_done_count, unsigned *flag, unsigned seq) {
  *h_done_count = *n_done;
  __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

hipError_t launch_chain_marker(const unsigned *n_done, unsigned *h_done_count, unsigned *flag, unsigned seq,
                               hipStream_t stream) {
  hipLaunchKernelGGL(k_chain_marker, dim3(1), dim3(1), 0, stream, n_done, h_done_count, flag, seq);
  return hipGetLastError();
}

hipError_t launch_hc_chain_step(const HcChainArgs &a, int cell_model, int k, int nt, hipStream_t stream,
                                hipEvent_t e0, hipEvent_t e1, int n_chains) {
  if (a.jobs) {  // a batch of independent matches: the default (canonical-sum) mode of the 1-cell OOPE
    if (a.seq) return hipErrorInvalidValue;
    if (cell_model == SLAMHIP_CELL_OCC) return launch_nt<SLAMHIP_CELL_OCC, false, true>(a, k, nt, stream, e0, e1, n_chains);
    if (cell_model == SLAMHIP_CELL_TBM) return launch_nt<SLAMHIP_CELL_TBM, false, true>(a, k, nt, stream, e0, e1, n_chains);
    return hipErrorInvalidValue;
  }
  if (cell_model == SLAMHIP_CELL_OCC)
    return a.seq ? launch_nt<SLAMHIP_CELL_OCC, true, false>(a, k, nt, stream, e0, e1, n_chains)
                 : launch_nt<SLAMHIP_CELL_OCC, false, false>(a, k, nt, stream, e0, e1, n_chains);
  if (cell_model == SLAMHIP_CELL_TBM)
    return a.seq ? launch_nt<SLAMHIP_CELL_TBM, true, false>(a, k, nt, stream, e0, e1, n_chains)
                 : launch_nt<SLAMHIP_CELL_TBM, false, false>(a, k, nt, stream, e0, e1, n_chains);
  if (cell_model == SLAMHIP_CELL_GMAPPING && !a.seq) {
    switch ((a.scan.n + 255) / 256) {
      case 1: return launch_gm<1>(a, k, nt, stream, e0, e1, n_chains);
      case 2: return launch_gm<2>(a, k, nt, stream, e0, e1, n_chains);
      case 3: return launch_gm<3>(a, k, nt, stream, e0, e1, n_chains);
      case 4: return launch_gm<4>(a, k, nt, stream, e0, e1, n_chains);
      case 5: return launch_gm<5>(a, k, nt, stream, e0, e1, n_chains);
      default: break;
    }
  }
  return hipErrorInvalidValue;
}

}  // namespace slamhip
