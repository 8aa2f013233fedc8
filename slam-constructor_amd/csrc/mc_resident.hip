// mc_resident.hip -- one Monte-Carlo process_scan as ONE launch of co-resident workgroups: mc_chain.hip's chain of
// kernels with the kernel boundary and the staging of the scores through memory taken out, the way hc_resident.hip
// does it for the hill-climbing matcher (design, visibility argument and the measurements behind it: there).
//
//   PoseEnumerationScanMatcher::process_scan      src/core/scan_matchers/pose_enumeration_scan_matcher.h:31-77
//   MonteCarloScanMatcher + GaussianPoseEnumerator src/core/scan_matchers/monte_carlo_scan_matcher.h:10-100
//
// The n_slots + 1 one-pose workgroups (at most 511 candidates + the bookkeeping workgroup, 512 threads each, two on a CU:
// 509 + 1 on an MI355X, where the launch leaves one CU's worth of workgroups spare)
// are launched ONCE and loop over the super-steps.  A workgroup scores its candidate of the current state (mc_chain.h's
// closed forms: every candidate hangs off the same best pose under "all rejected so far"), publishes {score,
// fingerprint, tag} as one 16-byte write-through granule, and its wave 0 gathers the granules of the candidates the
// state still hands out, finds the first accepted one (lane l looks at candidates 8 l .. 8 l + 7: mc_chain.hip's
// replay, the same decisions bit for bit) and advances the state.  Every spin is bounded: a sweep that does not
// complete stores the match's epoch in HcResidentCtl::fail_epoch, reports error 4 and leaves; the host then runs the
// match as the chain of kernels.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "hc_resident_device.h"
#include "mc_chain_device.h"
#include "score_device.h"

namespace slamhip {

namespace {
constexpr int kSumLanes = 256;
constexpr int kMcGran = kMcPerLane;  // granules per lane of the sweeping wave: 512 slots
}  // namespace

template <int MODEL, int NT, bool SEQ>
__global__ __launch_bounds__(NT, 4) void k_mc_chain_resident(McChainArgs a) {
  extern __shared__ double s_term[];  // one term per beam; with lds_consts: range, cosine, sine of the beams >= NT
  __shared__ double s_sc[kMcSlots + 8];
  __shared__ unsigned long long s_hash[kMcSlots + 8];  // (48 bits each)
  __shared__ McState s_st;  // the state the super-step about to be scored starts from
  __shared__ double s_pose[2][4];
  __shared__ int s_go[2], s_mode[2];
  __shared__ int s_stop;
  __shared__ double s_part[4];
  __shared__ unsigned long long s_hpart[4];
  const int t0 = threadIdx.x, wave = t0 >> 6;
  const bool init_slot = blockIdx.x + 1 == gridDim.x;  // the bookkeeping workgroup: slot kMcSlots
  const int slot = init_slot ? kMcSlots : (int)blockIdx.x;
  McResidentCtl *const rc = a.rctl;
  McHostOut *const host = a.host;
  // (a workgroup that starts after the others gave up leaves at once; ONE thread looks and the workgroup decides
  // behind a barrier -- every thread for itself could let some waves of a workgroup leave and others stay: ADVICE r4)
  // (the word is a trip to memory: asked for here, looked at below once this thread's beam is on its way too)
  const unsigned fail_epoch_at_entry = __hip_atomic_load(&rc->fail_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef SLAMHIP_TESTING
  if (a.debug_mute && (int)blockIdx.x + 1 == a.debug_mute) return;  // (the others must give up, not hang)
#endif
  const __attribute__((address_space(4))) McChainArgs *ap0 =
      (const __attribute__((address_space(4))) McChainArgs *)__builtin_amdgcn_kernarg_segment_ptr();
  const ScanView scan = load_view(&ap0->scan);
  const int n = scan.n;
  double br = 0.0, bc = 0.0, bs = 0.0, bw = 0.0, bf = 0.0;
  if (t0 < n) {
    br = scan.range[t0];
    bc = scan.cos_a[t0];
    bs = scan.sin_a[t0];
    bw = scan.weight[t0];
    bf = scan.factor[t0];
  }
  // (what the cell address of a thread's further beams depends on stays in LDS for the match: hc_resident.hip)
  const bool ldsc = a.lds_consts != 0;
  const int n_more = n > NT ? n - NT : 0;
  double *const s_r = s_term + n - NT, *const s_ca = s_r + n_more, *const s_sa = s_ca + n_more;  // (indexed by beam >= NT)
  if (ldsc) {
    for (int b = NT + t0; b < n; b += NT) {
      s_r[b] = scan.range[b];
      s_ca[b] = scan.cos_a[b];
      s_sa[b] = scan.sin_a[b];
    }
  }
  if (t0 == 0) s_stop = fail_epoch_at_entry == a.epoch ? 1 : 0;
  __syncthreads();
  if (s_stop) return;  // started after the others gave up (uniform: thread 0's reading)
  if (t0 < 4) {  // this slot's granules of both parities start the match empty (hc_tag)
    HcGranule *g0 = (t0 & 2) ? &rc->seq[t0 & 1][slot] : &rc->gran[t0 & 1][slot];
    gran_store(g0, 0.0, 0ull, 0u);
  }
  const bool verify = a.verify != 0;
  const bool stamp = SLAMHIP_STAMPS_ON(a.stamps && slot == 1 && t0 == 0);
  HcGranule *const gran = &rc->gran[0][0];
  HcGranule *const gseq = &rc->seq[0][0];
  constexpr int kGranRow = kMcSlots + 1;
  if (t0 == 0) {
    McState st{};
    st.x = a.init[0];
    st.y = a.init[1];
    st.theta = a.init[2];
    st.td = a.td0;
    st.rd = a.rd0;
    st.first = 1;
    s_st = st;
  }
  __syncthreads();

  for (int k = 0;; ++k) {
    const int pk = k & 1;
    // (the thread index and the kernel arguments as values the compiler cannot see through: nothing derived from them
    // is hoisted out of this loop and held in registers across it -- hc_resident.hip)
    int t = t0;
    asm volatile("" : "+v"(t));
    const int lane = t & 63;
    const __attribute__((address_space(4))) McChainArgs *ap =
        (const __attribute__((address_space(4))) McChainArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ap));
    const MapView map = load_view(&ap->map);
    // ---- wave 0: this workgroup's pose of super-step k
    if (wave == 0) {
      if (stamp && k < 64) ap->stamps[8 * k + 0] = wall_clock64();
      const McState &st = s_st;
      bool go = !st.done;
      double px = st.x, py = st.y, pth = st.theta;
      if (init_slot) {
        go = go && (st.first || st.mode == 1);  // the initial pose / the best pose of a re-scored super-step
      } else if (go) {
        const int avail = (int)mc_available(st, ap->max_failed, ap->max_poses);
        go = slot < (avail < ap->n_slots ? avail : ap->n_slots);
        if (go) mc_candidate(st, ap->tape, slot, &px, &py, &pth);
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
        s_mode[pk] = st.mode;
        if (st.done) s_stop = 1;
      }
      if (stamp && k < 64) ap->stamps[8 * k + 3] = wall_clock64();
    }
    __syncthreads();  // (A)
    if (s_stop) break;
    const int go = s_go[pk], mode = s_mode[pk];
    const unsigned tag = hc_tag(ap->tag_epoch, k);
    if (go) {
      const double px = s_pose[pk][0], py = s_pose[pk][1], sn = s_pose[pk][2], cs = s_pose[pk][3];
      // ---- score it: mc_chain.hip's body (terms by beam, the gathers of a round in flight together, canonical sum)
      constexpr int UNR = 4;
      for (int base = t; base < n; base += UNR * NT) {
        double4 cell[UNR];
        double w_[UNR], f_[UNR];
#pragma unroll
        for (int j = 0; j < UNR; ++j) {
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
            r_ = ldsc ? s_r[bc_] : scan.range[bc_];
            ca = ldsc ? s_ca[bc_] : scan.cos_a[bc_];
            sa = ldsc ? s_sa[bc_] : scan.sin_a[bc_];
            w_[j] = scan.weight[bc_];
            f_[j] = scan.factor[bc_];
          }
          cell[j] = beam_cell<MODEL>(map, px, py, sn, cs, r_, ca, sa);
        }
#pragma unroll
        for (int j = 0; j < UNR; ++j) {
          const int b = base + j * NT;
          if (b < n) s_term[b] = cell_probability<MODEL>(ap->oie, cell[j]) * w_[j] * f_[j];
        }
      }
      __syncthreads();  // (B)
      if (stamp && k < 64) ap->stamps[8 * k + 4] = wall_clock64();
      if (SEQ) {
        if (t == 0) {
          double acc = 0.0;
          for (int b = 0; b < n; ++b) acc = acc + s_term[b];
          gran_store(&gran[pk * kGranRow + slot], (scan.tot_w == 0.0) ? __builtin_nan("") : acc / scan.tot_w, 0ull, tag);
        }
        __syncthreads();  // (C) (s_term is rewritten by the next super-step)
      } else {
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
        __syncthreads();  // (C)
        if (t == 0) {
          const double total = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
          const unsigned long long fp = verify ? fold_fingerprint48(s_hpart[0] + s_hpart[1] + s_hpart[2] + s_hpart[3]) : 0ull;
          gran_store(&gran[pk * kGranRow + slot], (scan.tot_w == 0.0) ? __builtin_nan("") : total / scan.tot_w, fp, tag);
        }
        if (verify && mode && t == 64) {
          // re-scored super-step: the reference's own order as well, one running sum over the beams
          double acc = 0.0;
          for (int b = 0; b < n; ++b) acc = acc + s_term[b];
          gran_store(&gseq[pk * kGranRow + slot], (scan.tot_w == 0.0) ? __builtin_nan("") : acc / scan.tot_w, 0ull, tag);
        }
      }
      if (stamp && k < 64) ap->stamps[8 * k + 5] = wall_clock64();
    } else if (t == 0) {
      // nothing to score: the tag goes out all the same.  The sweepers wait for EVERY workgroup of the grid in every
      // super-step, so nobody -- the bookkeeping workgroup streaming an observer's trace over PCIe least of all -- is
      // ever more than one super-step behind the others, whose next-but-one granules would overwrite what it still
      // has to read
      gran_store(&gran[pk * kGranRow + slot], 0.0, 0ull, tag);
    }

    // ---- wave 0: the granules of the whole grid, then mc_chain.hip's replay over the candidates the state hands out
    if (wave == 0) {
      const McState &sp = s_st;
      const int avail = (int)mc_available(sp, ap->max_failed, ap->max_poses);
      const int n_cand = avail < ap->n_slots ? avail : ap->n_slots;
      const bool base_here = sp.first || sp.mode == 1;
      const int n_grid = ap->n_slots;  // (+ the bookkeeping workgroup)
      const bool rescored = !SEQ && verify && sp.mode == 1;  // decisions from the beam-order sums of this super-step
      bool failed = false;
      {
        const HcGranule *g0 = gran + pk * kGranRow;
        unsigned spins = 0;
        for (;;) {
          u32x4 g[kMcGran];
          const HcGranule *gp[kMcGran];
          bool ok = true;
#pragma unroll
          for (int q = 0; q < kMcGran; ++q) {
            const int i = lane + 64 * q;
            gp[q] = g0 + (i < n_grid ? i : kMcSlots);  // (behind the grid: the bookkeeping workgroup's, once more)
          }
          gran_fetch(g, gp);
#pragma unroll
          for (int q = 0; q < kMcGran; ++q) {
            const int i = lane + 64 * q;
            const int j = i < n_grid ? i : kMcSlots;
            const bool here = gran_tag(g[q]) == tag;
            ok = ok && here;
            if (here) {
              s_sc[j] = gran_score(g[q]);
              s_hash[j] = gran_hash(g[q]);
            }
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
      }
      if (stamp && k < 64) ap->stamps[8 * k + 1] = wall_clock64();
      if (failed || k + 1 >= kHcResidentMaxSteps) {
        // a workgroup of the grid is not on the chip (or the chain is longer than a tag can count): everybody leaves,
        // the host runs the match as the chain of kernels
        if (lane == 0) {
          __hip_atomic_store(&rc->fail_epoch, ap->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&host->error, failed ? 4 : 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __threadfence_system();
          __hip_atomic_store(&host->done_seq, ap->epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
          s_stop = 1;
        }
        continue;  // to (A), where the workgroup leaves
      }
      // ---- replay: lane l looks at candidates 8 l .. 8 l + 7 (mc_chain.hip)
      const double root = sp.first ? s_sc[kMcSlots] : sp.best_prob;
      const unsigned long long root_hash = verify ? (base_here ? s_hash[kMcSlots] : sp.best_hash) : 0ull;
      double root_dec = root;
      double dec6[kMcPerLane];
#pragma unroll
      for (int c = 0; c < kMcPerLane; ++c) {
        const int j = kMcPerLane * lane + c;
        dec6[c] = s_sc[j < kMcSlots ? j : 0];
      }
      if (rescored) {
        // the beam-order sums were stored next to the canonical ones by other lanes: wait for their tags as well, one
        // granule at a time (the rare super-step: a rolled loop that costs no registers)
        const HcGranule *q0 = gseq + pk * kGranRow;
#pragma unroll 1
        for (int c = -1; c < kMcPerLane; ++c) {
          const int j = c < 0 ? kMcSlots : kMcPerLane * lane + c;
          const bool live = c < 0 || j < n_cand;
          double sd = 0.0;
          for (unsigned spins = 0;; ++spins) {
            u32x4 g[1];
            const HcGranule *gp[1] = {q0 + (live ? j : kMcSlots)};
            gran_fetch(g, gp);
            sd = gran_score(g[0]);
            if (__all(gran_tag(g[0]) == tag) || spins > ap->spin_limit) break;  // (cannot run out: the canonical granules of
          }                                                               // the same workgroups are here already)
          if (c < 0) root_dec = sd;
          else dec6[c] = live ? sd : 0.0;
        }
      }
      int first_c = kMcPerLane;  // this lane's first accepted candidate
      unsigned amb_mask = 0u;  // candidates of this lane whose comparison the tree sum cannot settle
#pragma unroll
      for (int c = kMcPerLane - 1; c >= 0; --c) {
        const int j = kMcPerLane * lane + c;
        const bool live = j < n_cand;
        const double s = s_sc[j < kMcSlots ? j : 0];
        const double d = dec6[c];
        if (live && root_dec < d) first_c = c;  // strict: ties are rejections (pose_enumeration_scan_matcher.h:58)
        if (verify && !rescored && live) {
          const double diff = __builtin_fabs(s - root);
          const double as = __builtin_fabs(s), ab = __builtin_fabs(root);
          // (equal fingerprints with different sums: not identical vectors -- a collision, equally unsettled)
          const int close = (int)(diff <= (as > ab ? as : ab) * 9.094947017729282e-13);
          const int differ = (int)(s_hash[j] != root_hash) | (int)(__double_as_longlong(s) != __double_as_longlong(root));
          amb_mask |= (unsigned)(close & differ) << c;
        }
      }
      const unsigned long long acc_lanes = __ballot(first_c < kMcPerLane);
      const int acc_lane = acc_lanes ? __ffsll((long long)acc_lanes) - 1 : -1;
      const int j_acc = acc_lane < 0 ? -1 : kMcPerLane * acc_lane + __builtin_amdgcn_readlane(first_c, acc_lane < 0 ? 0 : acc_lane);
      const int used = j_acc >= 0 ? j_acc + 1 : n_cand;  // scorer calls of this super-step, in order
      unsigned mine = amb_mask;  // an unsettled comparison among the calls that count?
      if (kMcPerLane * lane + kMcPerLane - 1 >= used) {
        const int keep = used - kMcPerLane * lane;  // candidates of this lane below `used`
        mine = keep <= 0 ? 0u : (amb_mask & ((1u << keep) - 1u));
      }
      const bool dirty = !SEQ && verify && !rescored && __ballot(mine != 0u) != 0ull;
      if (stamp && k < 64) ap->stamps[8 * k + 7] = wall_clock64();
      McState next = sp;
      if (!dirty) {
        double ax = 0.0, ay = 0.0, ath = 0.0;
        if (j_acc >= 0) mc_candidate(sp, ap->tape, j_acc, &ax, &ay, &ath);
        const double aprob = j_acc >= 0 ? s_sc[j_acc] : 0.0;
        const unsigned long long ahash = (verify && j_acc >= 0) ? s_hash[j_acc] : 0ull;
        if (sp.first) {  // the initial pose was scored in the same super-step: call number one
          next.best_prob = root;
          next.best_hash = root_hash;
          next.calls = 1;
        }
        mc_advance(next, ap->tape, n_cand, j_acc, ax, ay, ath, aprob, ahash, ap->max_failed, ap->max_poses);
        next.evaluated = sp.evaluated + n_cand + (sp.first ? 1 : 0);
        next.first = 0;
        next.mode = 0;
      } else {
        next.mode = 1;  // same state, same candidates, once more with the beam-order sum as well
        next.evaluated = sp.evaluated + n_cand + 1;
        next.rescored = sp.rescored + 1;
      }
      next.steps = sp.steps + 1;
      if (stamp && k < 64) ap->stamps[8 * k + 2] = wall_clock64();
      if (init_slot) {
        // ---- the last workgroup keeps the books (it scores nothing after the first super-step)
        if (ap->trace && !dirty) {
          McTraceEntry *const trace = ap->trace;
          const long long base = sp.calls + (sp.first ? 1 : 0);
          if (sp.first && lane == 0 && ap->trace_cap > 0) {
            McTraceEntry e{sp.x, sp.y, sp.theta, root, 1, 0};
            trace[0] = e;
          }
          for (int c = 0; c < kMcPerLane; ++c) {
            const int j = kMcPerLane * lane + c;
            if (j < used) {
              McTraceEntry e;
              mc_candidate(sp, ap->tape, j, &e.x, &e.y, &e.theta);
              e.score = s_sc[j];
              e.accepted = j == j_acc ? 1 : 0;
              e.pad = 0;
              const long long at = base + j;
              if (at < ap->trace_cap) trace[at] = e;
              else host->error = 2;
            }
          }
        }
        if (next.done) {
          __threadfence_system();  // every lane's trace stores first, then the result, then the flag the host spins on
          if (lane == 0) {
            McHostOut *h = host;
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
            __hip_atomic_store(&h->done_seq, ap->epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
          }
        }
      }
      if (lane == 0) {
        // (`sp` above is this very object: the books are kept from the old state first; field by field -- the struct
        // assigned as a whole may travel through scratch, hc_resident.hip)
        McState &w = s_st;
        w.x = next.x;
        w.y = next.y;
        w.theta = next.theta;
        w.best_prob = next.best_prob;
        w.td = next.td;
        w.rd = next.rd;
        w.saved[0] = next.saved[0];
        w.saved[1] = next.saved[1];
        w.saved[2] = next.saved[2];
        w.pos = next.pos;
        w.calls = next.calls;
        w.evaluated = next.evaluated;
        w.failed = next.failed;
        w.poses = next.poses;
        w.has_saved = next.has_saved;
        w.done = next.done;
        w.first = next.first;
        w.mode = next.mode;
        w.steps = next.steps;
        w.best_hash = next.best_hash;
        w.rescored = next.rescored;
      }
    }
  }
}

// dynamic LDS of a workgroup (the terms, and with lds_consts the further beams' range / cosine / sine)
size_t mc_resident_lds_bytes(int nt, int n_beams, bool lds_consts) {
  const size_t n = (size_t)(n_beams > 0 ? n_beams : 1);
  const size_t more = lds_consts && n > (size_t)nt ? n - (size_t)nt : 0;
  return sizeof(double) * (n + 3 * more);
}

#define MCR_LAUNCH(NTV)                                                                                            \
  do {                                                                                                             \
    if (e0 || e1)                                                                                                  \
      hipExtLaunchKernelGGL((k_mc_chain_resident<MODEL, NTV, SEQ>), dim3(grid), dim3(NTV), shm, stream, e0, e1, 0, a); \
    else                                                                                                           \
      hipLaunchKernelGGL((k_mc_chain_resident<MODEL, NTV, SEQ>), dim3(grid), dim3(NTV), shm, stream, a);            \
  } while (0)

template <int MODEL, bool SEQ>
static hipError_t launch_mcr(const McChainArgs &a, int nt, hipStream_t stream, hipEvent_t e0, hipEvent_t e1) {
  const int grid = a.n_slots + 1;
  const size_t shm = mc_resident_lds_bytes(nt, a.scan.n, a.lds_consts != 0);
  if (nt == 512) MCR_LAUNCH(512);
  else MCR_LAUNCH(1024);
  return hipGetLastError();
}
#undef MCR_LAUNCH

hipError_t launch_mc_chain_resident(const McChainArgs &a, int cell_model, int nt, hipStream_t stream, hipEvent_t e0,
                                    hipEvent_t e1) {
  if (!a.rctl || a.n_slots < 1 || a.n_slots > kMcSlots) return hipErrorInvalidValue;
  if (cell_model == SLAMHIP_CELL_OCC)
    return a.seq ? launch_mcr<SLAMHIP_CELL_OCC, true>(a, nt, stream, e0, e1)
                 : launch_mcr<SLAMHIP_CELL_OCC, false>(a, nt, stream, e0, e1);
  if (cell_model == SLAMHIP_CELL_TBM)
    return a.seq ? launch_mcr<SLAMHIP_CELL_TBM, true>(a, nt, stream, e0, e1)
                 : launch_mcr<SLAMHIP_CELL_TBM, false>(a, nt, stream, e0, e1);
  return hipErrorInvalidValue;
}

// workgroups of `nt` threads the device keeps resident at once (hc_resident_capacity's rule)
hipError_t mc_resident_capacity(int cell_model, int nt, int n_beams, bool lds_consts, int *out_wgs, int *out_per_cu) {
  int dev = 0, cus = 0, per_cu = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (e != hipSuccess) return e;
  const void *fn = nullptr;
  if (cell_model == SLAMHIP_CELL_TBM)
    fn = nt == 1024 ? (const void *)k_mc_chain_resident<SLAMHIP_CELL_TBM, 1024, false>
                    : (const void *)k_mc_chain_resident<SLAMHIP_CELL_TBM, 512, false>;
  else
    fn = nt == 1024 ? (const void *)k_mc_chain_resident<SLAMHIP_CELL_OCC, 1024, false>
                    : (const void *)k_mc_chain_resident<SLAMHIP_CELL_OCC, 512, false>;
  e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, nt, mc_resident_lds_bytes(nt, n_beams, lds_consts));
  if (e != hipSuccess) return e;
  const int by_waves = 2048 / nt;  // 128-VGPR waves: four per SIMD
  per_cu = per_cu < by_waves ? per_cu : by_waves;
  *out_wgs = per_cu * (cus - 1);  // (one CU's worth of margin: hc_resident_capacity)
  if (out_per_cu) *out_per_cu = per_cu;
  return hipSuccess;
}

}  // namespace slamhip
