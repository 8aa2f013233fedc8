// mc_chain.hip -- one Monte-Carlo process_scan as a chain of kernels with no host in between (state and closed
// forms: mc_chain.h; the hill-climbing counterpart and the design notes: hc_chain.hip).
//
// Kernel k_mc_chain_step, launched back to back on the context's stream with k = 0, 1, 2, ...:
//   prologue (k > 0)  replay of super-step k-1 by wave 0 of EVERY workgroup: the scores of the candidates the
//                     previous super-step speculated on ("all of them rejected", every one hanging off the same
//                     best pose), first `best < candidate` wins (strict, the reference's order), everything behind
//                     it is discarded; the enumerator state after the consumed candidates is a closed form.  The
//                     last workgroup also stores the state, writes the observer trace and publishes the result.
//   body              workgroup w scores candidate w of the new state (the last workgroup: the initial pose in the
//                     first super-step, the best pose when a super-step is scored twice) with k_score_point's
//                     arithmetic and canonical sum, so its bits equal the host-driven matcher's.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "mc_chain_device.h"
#include "score_device.h"

namespace slamhip {

static constexpr int kSumLanes = 256;
#define MC_PIN32(x) asm volatile("" ::"s"(x))
#define MC_PIN64(x) asm volatile("" ::"s"((unsigned long long)(x)))
#define MC_PINF(x) asm volatile("" ::"s"(__double_as_longlong(x)))

template <int MODEL, int NT, bool SEQ>
__global__ __launch_bounds__(NT) void k_mc_chain_step(McChainArgs a, int k) {
  extern __shared__ double s_term[];
  __shared__ double s_sc[kMcSlots + 8];
  __shared__ unsigned long long s_hash[kMcSlots + 8];
  __shared__ McState s_prev;
  __shared__ double s_pose[4];
  __shared__ int s_go, s_mode;
  __shared__ double s_part[4];
  __shared__ unsigned long long s_hpart[4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const bool init_slot = blockIdx.x + 1 == gridDim.x;  // the bookkeeping workgroup: slot kMcSlots
  const int slot = init_slot ? kMcSlots : (int)blockIdx.x;
  McChainCtl *ctl = a.ctl;
  const unsigned done_epoch = __hip_atomic_load(&ctl->done_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int n = a.scan.n;
  double br = 0.0, bc = 0.0, bs = 0.0, bw = 0.0, bf = 0.0;
  if (t < n) {
    br = a.scan.range[t];
    bc = a.scan.cos_a[t];
    bs = a.scan.sin_a[t];
    bw = a.scan.weight[t];
    bf = a.scan.factor[t];
  }
  MC_PIN32(a.max_failed);
  MC_PIN32(a.max_poses);
  MC_PIN32(a.n_slots);
  MC_PIN64(a.tape);
  MC_PIN64(a.host);
  MC_PIN64(a.trace);
  MC_PIN32(a.trace_cap);
  MC_PIN64(a.map.payload);
  MC_PIN32(a.map.width);
  MC_PIN32(a.map.height);
  MC_PIN32(a.map.pitch);
  MC_PIN32(a.map.origin_x);
  MC_PIN32(a.map.origin_y);
  MC_PINF(a.map.scale);
  MC_PINF(a.map.inv_scale);
  MC_PINF(a.map.unknown[0]);
  MC_PINF(a.map.unknown[1]);
  MC_PINF(a.map.unknown[2]);
  MC_PINF(a.map.unknown[3]);
  MC_PIN32(a.oie);
  MC_PINF(a.scan.tot_w);
  const int pb = (k - 1) & 1;
  if (k > 0) {
    const double *sc_prev = ctl->scores[pb];
    for (int i = t; i <= a.n_slots; i += NT) {
      const int j = i < a.n_slots ? i : kMcSlots;
      s_sc[j] = sc_prev[j];
      if (a.verify) s_hash[j] = ctl->hashes[pb][j];
    }
    if (t < (int)(sizeof(McState) / 8))
      reinterpret_cast<double *>(&s_prev)[t] = reinterpret_cast<const double *>(&ctl->state[pb])[t];
  }
  if (done_epoch == a.epoch) return;  // launched past the end of the chain (uniform: before any barrier)
  __syncthreads();

  if (wave == 0) {
    McState st;
    if (k == 0) {
      st = McState{};
      st.x = a.init[0];
      st.y = a.init[1];
      st.theta = a.init[2];
      st.td = a.td0;
      st.rd = a.rd0;
      st.first = 1;
      if (init_slot && lane == 0) {
        ctl->state[0] = st;
        __hip_atomic_store(&a.host->progress, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    } else {
      // ---- replay of the previous super-step: lane l looks at candidates 8 l .. 8 l + 7 (kMcPerLane)
      const McState sp = s_prev;
      const bool verify = a.verify != 0;
      const bool rescored = verify && sp.mode == 1;  // decisions from the beam-order sums of this super-step
      const double root = sp.first ? s_sc[kMcSlots] : sp.best_prob;
      const bool base_here = sp.first || sp.mode == 1;
      const unsigned long long root_hash = verify ? (base_here ? s_hash[kMcSlots] : sp.best_hash) : 0ull;
      const double *seq = ctl->scores_seq[pb];
      const double root_dec = rescored ? seq[kMcSlots] : root;
      const int avail = (int)mc_available(sp, a.max_failed, a.max_poses);
      const int n_cand = avail < a.n_slots ? avail : a.n_slots;
      int first_c = kMcPerLane;         // this lane's first accepted candidate
      unsigned amb_mask = 0u;  // candidates of this lane whose comparison the tree sum cannot settle
#pragma unroll
      for (int c = kMcPerLane - 1; c >= 0; --c) {
        const int j = kMcPerLane * lane + c;
        const bool live = j < n_cand;
        const double s = s_sc[j < kMcSlots ? j : 0];
        const double d = rescored ? (live ? seq[j] : 0.0) : s;
        if (live && root_dec < d) first_c = c;  // strict: ties are rejections (pose_enumeration_scan_matcher.h:58)
        if (verify && !rescored && live) {
          const double diff = __builtin_fabs(s - root);
          const double as = __builtin_fabs(s), ab = __builtin_fabs(root);
          // (equal fingerprints with different sums: not identical vectors -- a collision, equally unsettled)
          if (diff <= (as > ab ? as : ab) * 9.094947017729282e-13 &&
              (s_hash[j] != root_hash || __double_as_longlong(s) != __double_as_longlong(root)))
            amb_mask |= 1u << c;
        }
      }
      const unsigned long long acc_lanes = __ballot(first_c < kMcPerLane);
      const int acc_lane = acc_lanes ? __ffsll((long long)acc_lanes) - 1 : -1;
      const int j_acc = acc_lane < 0 ? -1 : kMcPerLane * acc_lane + __builtin_amdgcn_readlane(first_c, acc_lane < 0 ? 0 : acc_lane);
      const int used = j_acc >= 0 ? j_acc + 1 : n_cand;  // scorer calls of this super-step, in order
      // an unsettled comparison among the calls that count?
      unsigned mine = amb_mask;
      if (kMcPerLane * lane + kMcPerLane - 1 >= used) {
        const int keep = used - kMcPerLane * lane;  // candidates of this lane below `used`
        mine = keep <= 0 ? 0u : (amb_mask & ((1u << keep) - 1u));
      }
      const bool dirty = verify && !rescored && __ballot(mine != 0u) != 0ull;
      McState next = sp;
      double ax = 0.0, ay = 0.0, ath = 0.0;
      if (!dirty) {
        if (j_acc >= 0) mc_candidate(sp, a.tape, j_acc, &ax, &ay, &ath);
        const double aprob = j_acc >= 0 ? s_sc[j_acc] : 0.0;
        const unsigned long long ahash = (verify && j_acc >= 0) ? s_hash[j_acc] : 0ull;
        if (sp.first) {  // the initial pose was scored in the same super-step: call number one
          next.best_prob = root;
          next.best_hash = root_hash;
          next.calls = 1;
        }
        mc_advance(next, a.tape, n_cand, j_acc, ax, ay, ath, aprob, ahash, a.max_failed, a.max_poses);
        next.evaluated = sp.evaluated + n_cand + (sp.first ? 1 : 0);
        next.first = 0;
        next.mode = 0;
      } else {
        next.mode = 1;  // same state, same candidates, once more with the beam-order sum as well
        next.evaluated = sp.evaluated + n_cand + 1;
        next.rescored = sp.rescored + 1;
      }
      next.steps = sp.steps + 1;
      st = next;
      if (init_slot) {
        if (a.trace && !dirty) {
          const long long base = sp.calls + (sp.first ? 1 : 0);
          if (sp.first && lane == 0 && a.trace_cap > 0) {
            McTraceEntry e{sp.x, sp.y, sp.theta, root, 1, 0};
            a.trace[0] = e;
          }
          for (int c = 0; c < kMcPerLane; ++c) {
            const int j = kMcPerLane * lane + c;
            if (j < used) {
              McTraceEntry e;
              mc_candidate(sp, a.tape, j, &e.x, &e.y, &e.theta);
              e.score = s_sc[j];
              e.accepted = j == j_acc ? 1 : 0;
              e.pad = 0;
              const long long at = base + j;
              if (at < a.trace_cap) a.trace[at] = e;
              else a.host->error = 2;
            }
          }
        }
        if (lane == 0) ctl->state[k & 1] = next;
        if (next.done) {
          __threadfence_system();
          if (lane == 0) {
            ctl->done_epoch = a.epoch;
            McHostOut *h = a.host;
            h->pose[0] = next.x;
            h->pose[1] = next.y;
            h->pose[2] = next.theta;
            h->best_prob = next.best_prob;
            h->calls = next.calls;
            h->evaluated = next.evaluated;
            h->steps = next.steps;
            h->rescored = next.rescored;
            h->tape_pos = next.pos;
            h->failed = next.failed;
            h->poses = next.poses;
            h->td = next.td;
            h->rd = next.rd;
            h->has_saved = next.has_saved;
            h->saved[0] = next.saved[0];
            h->saved[1] = next.saved[1];
            h->saved[2] = next.saved[2];
            __hip_atomic_store(&h->done_seq, a.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
          }
        } else if (lane == 0) {
          __hip_atomic_store(&a.host->progress, (unsigned)(k + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
    // ---- this workgroup's pose
    bool go = !st.done;
    double px = st.x, py = st.y, pth = st.theta;
    if (init_slot) {
      go = go && (st.first || st.mode == 1);
    } else if (go) {
      const int avail = (int)mc_available(st, a.max_failed, a.max_poses);
      go = slot < (avail < a.n_slots ? avail : a.n_slots);
      if (go) mc_candidate(st, a.tape, slot, &px, &py, &pth);
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
  if (!s_go) return;
  const double px = s_pose[0], py = s_pose[1], sn = s_pose[2], cs = s_pose[3];
  // ---- score it: hc_chain.hip's body (terms by beam, four gathers in flight, canonical sum + fingerprint)
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
        r_ = a.scan.range[bc_];
        ca = a.scan.cos_a[bc_];
        sa = a.scan.sin_a[bc_];
        w_[j] = a.scan.weight[bc_];
        f_[j] = a.scan.factor[bc_];
      }
      cell[j] = beam_cell<MODEL>(a.map, px, py, sn, cs, r_, ca, sa);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int b = base + j * NT;
      if (b < n) s_term[b] = cell_probability<MODEL>(a.oie, cell[j]) * w_[j] * f_[j];
    }
  }
  __syncthreads();
  if (SEQ) {
    if (t == 0) {
      double acc = 0.0;
      for (int b = 0; b < n; ++b) acc = acc + s_term[b];
      ctl->scores[k & 1][slot] = (a.scan.tot_w == 0.0) ? __builtin_nan("") : acc / a.scan.tot_w;
    }
    return;
  }
  const bool verify = a.verify != 0;
  if (t < kSumLanes) {
    double acc = 0.0;
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
    ctl->scores[k & 1][slot] = (a.scan.tot_w == 0.0) ? __builtin_nan("") : total / a.scan.tot_w;
    if (verify) ctl->hashes[k & 1][slot] = s_hpart[0] + s_hpart[1] + s_hpart[2] + s_hpart[3];
  }
  if (verify && s_mode && t == 64) {
    double acc = 0.0;
    for (int b = 0; b < n; ++b) acc = acc + s_term[b];
    ctl->scores_seq[k & 1][slot] = (a.scan.tot_w == 0.0) ? __builtin_nan("") : acc / a.scan.tot_w;
  }
}

#define MC_LAUNCH(NTV)                                                                                             \
  do {                                                                                                             \
    if (e0 || e1)                                                                                                  \
      hipExtLaunchKernelGGL((k_mc_chain_step<MODEL, NTV, SEQ>), dim3(grid), dim3(NTV), shm, stream, e0, e1, 0, a, k); \
    else                                                                                                           \
      hipLaunchKernelGGL((k_mc_chain_step<MODEL, NTV, SEQ>), dim3(grid), dim3(NTV), shm, stream, a, k);            \
  } while (0)

template <int MODEL, bool SEQ>
static hipError_t launch_nt(const McChainArgs &a, int k, int nt, hipStream_t stream, hipEvent_t e0, hipEvent_t e1) {
  const int grid = a.n_slots + 1;
  const size_t shm = sizeof(double) * (size_t)(a.scan.n > 0 ? a.scan.n : 1);
  if (nt == 512) MC_LAUNCH(512);
  else MC_LAUNCH(1024);
  return hipGetLastError();
}
#undef MC_LAUNCH

hipError_t launch_mc_chain_step(const McChainArgs &a, int cell_model, int k, int nt, hipStream_t stream,
                                hipEvent_t e0, hipEvent_t e1) {
  if (a.n_slots < 1 || a.n_slots > kMcSlots) return hipErrorInvalidValue;
  if (cell_model == SLAMHIP_CELL_OCC)
    return a.seq ? launch_nt<SLAMHIP_CELL_OCC, true>(a, k, nt, stream, e0, e1)
                 : launch_nt<SLAMHIP_CELL_OCC, false>(a, k, nt, stream, e0, e1);
  if (cell_model == SLAMHIP_CELL_TBM)
    return a.seq ? launch_nt<SLAMHIP_CELL_TBM, true>(a, k, nt, stream, e0, e1)
                 : launch_nt<SLAMHIP_CELL_TBM, false>(a, k, nt, stream, e0, e1);
  return hipErrorInvalidValue;
}

}  // namespace slamhip
