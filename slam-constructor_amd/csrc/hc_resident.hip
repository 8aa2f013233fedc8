// hc_resident.hip -- one hill-climbing process_scan as ONE launch of co-resident workgroups.
//
//   PoseEnumerationScanMatcher::process_scan      src/core/scan_matchers/pose_enumeration_scan_matcher.h:31-77
//   Distorsion1DPoseEnumerator +
//   FailedRoundsLimitedPoseEnumerator (HC)        src/core/scan_matchers/hill_climbing_scan_matcher.h:10-126
//
// hc_chain.hip runs a match as a chain of kernels: a super-step scores one speculation tree, the next kernel's
// prologue stages the 253 scores from memory and replays the accept chain over them.  Its fixed costs per
// super-step -- kernel boundary 1.8 us, staging 1.6 us -- are what this file removes: the same 6 x instances + 1
// one-pose workgroups are launched ONCE and loop over the super-steps.  A workgroup
//   scores its pose (k_score_point's arithmetic and canonical sum: same bits as every other path),
//   publishes {score, fingerprint, tag} as ONE 16-byte write-through (sc0 sc1) store -- a granule, the data is
//     its own flag --,
//   and wave 0 re-reads the granules of the whole tree (sc1 loads, they bypass the CU's L1) until every tag is
//     this super-step's, replays the accept chain (hc_chain.h: lane = round instance, seven ballots) and derives
//     the pose it scores next.
// Nobody waits for a barrier or a flag, only for data; every workgroup replays (nothing is broadcast).
// Measured before it was built (tools/probes/coresident_probe.hip, profiles/r04_coresident_probe.txt): granule
// all-gather 2.0 us from publish to "all 253 here" against 1.7 (boundary) + 0.9 (staging) in the probe's model of
// the kernel chain; an arrival counter is 5.0 us, three 8-byte granules per score 3.5 us, replaying once and
// broadcasting the root no better than everybody replaying.
//
// Residency.  The loop only terminates if every workgroup of the grid is on the chip at the same time.  The grid is
// sized by the host from the occupancy query (hc_resident_capacity) -- hipLaunchCooperativeKernel would add the same
// check at +15-19 us per launch (MI355X_MICROARCH.md, coop-launch) -- and EVERY spin is bounded: a sweep that does
// not complete within kHcSpinLimit polls stores the match's epoch in HcResidentCtl::fail_epoch, reports error 4 to the
// host and leaves; the other workgroups see the word and leave; workgroups that start later leave at once.  The
// host then runs the match on the kernel chain.  A hang is impossible by construction.
//
// Visibility (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility", form R2): a
// granule is one naturally aligned 16-byte sc1 store by one lane and is read by 16-byte sc1 loads; it was never
// observed torn (the probe checks a hash of the score in every granule it reads: 0 of 3 x 10^8), and the tag word
// is the LAST dword of the granule.  Everything else a workgroup reads during the loop is read-only for the match
// (map, scan, shapes) or its own LDS.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "hc_chain_device.h"
#include "hc_resident_device.h"
#include "score_device.h"

namespace slamhip {

namespace {
constexpr int kSumLanes = 256;          // the canonical sum's partials (= k_score_point's block)
}  // namespace

// What a workgroup tabulates, while the scores of super-step k are on their way, for every (terminal round instance,
// outcome) pair the tree of super-step k allows: the root of the next tree (hc_next_core: no score enters it) and the
// pose THIS workgroup scores in it, sine and cosine included.  The replay then only picks an entry -- the serial
// "advance" and "pose" phases of r04's super-step (0.54 + 0.69 us of 6.0, one wave at work and fifteen watching) are
// done by the waves that used to watch.
template <int H>  // poses per workgroup: 1, or 2 (the PAIR form of the batch chains)
struct HcNextEntryT {
  double x, y, theta;  // base of the next tree's root round (HcNextCore)
  double p[H][4];      // x, y, sin, cos of the pose(s) this workgroup scores in the next tree
  unsigned counts;     // failed rounds (16 bits) | scorer calls of the walked path (12) | its accepted rounds (4)
  unsigned flags;      // shape of the next tree (3 bits) | done (bit 3) | pose h is scored (bit 4 + h) | the chain ends
                       // on an inert root (bit 6: hc_inert -- `done` is set as well)
  // (the steps follow from the failed rounds -- one exact halving each: hc_round_of --, the acceptance-rate estimate
  // from the two counts)
};
static_assert(sizeof(HcNextEntryT<1>) == 64 && sizeof(HcNextEntryT<2>) == 96, "HcNextEntry layout");
// (16 bits of failed rounds, 12 of scorer calls, 4 of accepted rounds -- what they have to hold: the device
// forms run matchers of at most 1000 failed rounds (hc_limit_on_device, matchers.cpp), a walked path makes at most
// 6 x 42 + 6 scorer calls and accepts in at most kHcMaxSeg + 1 rounds)
static_assert(kHcMaxSeg + 1 <= 15, "HcNextEntry::counts: accepted rounds of a path in 4 bits");
static_assert(6 * kHcMaxInst + 6 < 4096, "HcNextEntry::counts: scorer calls of a path in 12 bits");
__device__ __forceinline__ unsigned hc_entry_counts(const HcNextCore &c, int rounds_acc) {
  return (c.failed & 0xffffu) | (((unsigned)c.batch_calls & 0xfffu) << 16) | (((unsigned)rounds_acc & 0xfu) << 28);
}

// minima of two values over a wave (every lane ends with both)
__device__ __forceinline__ void wave_min2(double &a, double &b) {
#define HCR_MIN_STEP(OFF)                              \
  {                                                    \
    const double oa = lane_xor_f64<OFF>(a), ob = lane_xor_f64<OFF>(b); \
    a = a < oa ? a : oa;                               \
    b = b < ob ? b : ob;                               \
  }
  HCR_MIN_STEP(32) HCR_MIN_STEP(16) HCR_MIN_STEP(8) HCR_MIN_STEP(4) HCR_MIN_STEP(2) HCR_MIN_STEP(1)
#undef HCR_MIN_STEP
}

// s_sel: what the pose of the next super-step is read from
constexpr int kSelRescore = -1;  // the same tree once more (an unsettled comparison): the poses just scored
constexpr int kSelFirst = -2;    // the first super-step: the prologue's pose
constexpr int kSelStop = -3;     // leave: the replay found no terminal round (a bug, reported) or the chain gave up

// MODEL: SLAMHIP_CELL_OCC / _TBM (the 1-cell OOPE); SEQ: the reference's beam-order sum; BATCH: grid.y independent
// matches, each with its own map and scan (HcChainArgs::jobs); G: granules per lane of the sweeping wave, i.e. the
// grid has at most 64 G workgroups (2, 4 or 7); WIN: the window OOPEs (max / mean / overlap, K2's per-beam value) in
// place of the 1-cell one
// (four waves per SIMD whatever the workgroup size: 128 VGPRs, so that 4 x 256, 2 x 512 or 1 x 1024 threads share a CU)
// PAIR (batches): a workgroup of 512 threads scores TWO poses per super-step, 256 threads each, side by side -- and
// sweeps, replays and tabulates ONCE for the two.  At saturation the CUs are short of issue slots (LOG r05), and what
// every workgroup does for itself besides its terms -- polling the granules, the replay, the table of next poses -- is
// ~45 % on top of them; the pair halves that share, and its two poses share the beam constants in LDS, which then fit
// again beside the table (K = 8).
template <int MODEL, int NT, bool SEQ, bool BATCH, int G, bool WIN = false, bool PAIR = false>
__global__ __launch_bounds__(NT, 4) void k_hc_chain_resident(HcChainArgs a) {
  constexpr int H = PAIR ? 2 : 1;    // poses per workgroup
  constexpr int NTH = NT / H;        // threads per pose
  constexpr bool CERT = !WIN;        // the 1-cell OOPE: a beam's term is a function of its cell alone (see "certificate")
  typedef HcNextEntryT<H> HcNextEntry;
  extern __shared__ double s_term[];  // one term per beam; behind them, for workgroups narrower than the scan: range,
                                      // cosine, sine of every further beam; then the table of next poses
                                      // (hc_resident_lds_bytes)
  __shared__ unsigned long long s_hash[kHcSlots + 7];  // (48 bits each)
  __shared__ double s_sc[kHcSlots + 7];
  __shared__ HcInst s_mine[H][kHcShapes];  // this workgroup's round instance(s) in every shape
  __shared__ double s_cur[H][4];           // x, y, sin, cos of the pose being scored (a re-scored super-step reads it again)
  __shared__ int s_cur_go[H];
  __shared__ int s_sel;                 // where the next super-step's pose comes from: a table entry, or kSel*
  __shared__ int s_stop;
  __shared__ HcState s_st;              // root state of the super-step being scored
  __shared__ double s_part[H][4];
  __shared__ unsigned long long s_hpart[H][4];
  __shared__ double s_cert[2][16];  // the bookkeeping workgroup's certificate: per-wave minima (see "certificate" below)
  __shared__ int s_cert_end;        // the replay's verdict: the chain ends on a certified root
  const int t = threadIdx.x, wave = t >> 6;
  const int half = PAIR ? (t >= NTH ? 1 : 0) : 0;  // which of the workgroup's poses this thread works on
  const int tl = t - half * NTH, lwave = tl >> 6;  // ... and its place among that pose's threads
  // the last workgroup keeps the books (and scores the initial pose / a re-scored base: its first half)
  const bool init_slot = blockIdx.x + 1 == gridDim.x;
  const int slot = init_slot ? kHcSlots - 1 : (PAIR ? 2 * (int)blockIdx.x + half : (int)blockIdx.x);
  const int inst_of_slot = slot / 6, cand = slot - 6 * inst_of_slot;
  const bool exists = init_slot ? half == 0 : slot < 6 * a.max_inst;  // (the bookkeeping workgroup's second half: no slot)
  HcResidentCtl *const rc = a.rctl + blockIdx.y;
  HcHostOut *const host = a.host + blockIdx.y;
  // a workgroup that starts after the others have given up (it was not resident with them) leaves at once -- the word
  // is a trip to memory (agent scope), so it is only looked at below, once this thread's beam is on its way too
  const unsigned fail_epoch_at_entry = __hip_atomic_load(&rc->fail_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef SLAMHIP_TESTING
  if (a.debug_mute && (int)blockIdx.x + 1 == a.debug_mute) return;  // (the others must give up, not hang)
#endif
  // this chain's map and scan: kernel arguments, or -- a batch of matches -- its entry of the job table
  MapViewCP map_p;
  ScanViewCP scan_p;
  {
    const __attribute__((address_space(4))) HcChainArgs *ap0 =
        (const __attribute__((address_space(4))) HcChainArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    if (BATCH) {
      const __attribute__((address_space(4))) HcJobView *jv =
          (const __attribute__((address_space(4))) HcJobView *)(unsigned long long)(a.jobs + blockIdx.y);
      map_p = &jv->map;
      scan_p = &jv->scan;
    } else {
      map_p = &ap0->map;
      scan_p = &ap0->scan;
    }
  }
  const ScanView scan = load_view(scan_p);
  const int n = scan.n;
  // this thread's first beam: its constants stay in registers for the whole match
  double br = 0.0, bc = 0.0, bs = 0.0, bw = 0.0, bf = 0.0;
  if (tl < n) {
    br = scan.range[tl];
    bc = scan.cos_a[tl];
    bs = scan.sin_a[tl];
    bw = scan.weight[tl];
    bf = scan.factor[tl];
  }
  if (lwave == 1 && (t & 63) < kHcShapes && !init_slot && exists) {
    const uint4 *src = reinterpret_cast<const uint4 *>(&a.shapes[t & 63].inst[inst_of_slot]);
    uint4 *dst = reinterpret_cast<uint4 *>(&s_mine[half][t & 63]);
#pragma unroll
    for (int q = 0; q < (int)(sizeof(HcInst) / 16); ++q) dst[q] = src[q];
  }
  // A thread scores up to five beams, and 1080 beams on 1024 threads give the first 56 threads a second one.  What
  // the cell ADDRESS of those further beams depends on -- range, cosine, sine -- is kept in LDS for the whole match
  // (the host asks for it when the workgroups still fit the device with that much LDS, HcChainArgs::lds_consts): read
  // from memory, as weight and factor still are, it was a round trip in front of the gathers' own in every
  // super-step (r04, super-step of a lone chain, us: 256 threads 7.3 -> 6.6, 512 6.3 -> 6.1).
  const bool ldsc = !WIN && a.lds_consts != 0;
  const int n_more = n > NTH ? n - NTH : 0;
  // (a pair: two term vectors, a.scan.n -- the launch's longest scan -- apart; the constants behind them are shared)
  double *const my_term = s_term + (PAIR ? half * a.scan.n : 0);
  double *const s_r = s_term + (PAIR ? 2 * a.scan.n : n) - NTH, *const s_ca = s_r + n_more, *const s_sa = s_ca + n_more;  // (indexed by beam >= NTH)
  if (ldsc) {
    for (int b = NTH + t; b < n; b += NT) {
      s_r[b] = scan.range[b];
      s_ca[b] = scan.cos_a[b];
      s_sa[b] = scan.sin_a[b];
    }
  }
  // (behind the terms and the beam constants of the LONGEST scan of the launch: hc_resident_lds_bytes)
  HcNextEntry *const s_tab = reinterpret_cast<HcNextEntry *>(s_term + a.tab_offset);
  // (ONE thread's reading decides for the workgroup, behind the barrier below: every thread for itself could let some
  // waves of a workgroup leave and others stay -- ADVICE r4)
  if (t == 0) s_stop = fail_epoch_at_entry == a.epoch ? 1 : 0;
  if (tl < 4 && exists) {  // this slot's granules of both parities start the match empty (see hc_tag)
    HcGranule *g0 = (tl & 2) ? &rc->seq[tl & 1][slot] : &rc->gran[tl & 1][slot];
    gran_store(g0, 0.0, 0ull, 0u);
  }
  const bool verify = a.verify != 0;
  const bool stamp = SLAMHIP_STAMPS_ON(a.stamps && slot == 1 && tl == 0 && blockIdx.y == 0);
  HcGranule *const gran = &rc->gran[0][0];
  HcGranule *const gseq = &rc->seq[0][0];
  constexpr int kGranRow = kHcSlots + 7;

  if (t == 0) {
    HcState st{};
    st.x = a.inits ? a.inits[3 * blockIdx.y] : a.init[0];
    st.y = a.inits ? a.inits[3 * blockIdx.y + 1] : a.init[1];
    st.theta = a.inits ? a.inits[3 * blockIdx.y + 2] : a.init[2];
    st.dt = a.dt0;
    st.dr = a.dr0;
    st.shape = a.shape0;
    st.first = 1;
    st.carry_cx = st.carry_cy = -1;
    st.carry_prob = -1.0;
    s_st = st;
    s_sel = kSelFirst;
    s_cert_end = 0;
  }
  __syncthreads();  // s_mine, s_stop, s_st
  if (s_stop) return;  // started after the others gave up (uniform)
  // ---- the first super-step's pose: the initial pose (bookkeeping workgroup), or this workgroup's candidate of the
  // first tree -- the one pose of a match that is derived serially
  if (lwave == 0) {
    const HcState &st = s_st;
    bool go = exists;
    double px = st.x, py = st.y, pth = st.theta;
    if (!init_slot && exists) {
      HcInst in;
#pragma unroll
      for (int q = 0; q < 14; ++q) in.w[q] = s_mine[half][st.shape].w[q];
      go = inst_of_slot < (int)((a.n_inst >> (8 * st.shape)) & 0xffull) &&
           (hc_is_root(in) || st.failed + hc_nfail_parent(in) < a.max_failed);  // else: behind the end of the chain
      if (go) {
        const HcRound r = hc_round_of(st, in);
        go = !(hc_trailing(r.failed, a.max_failed) && cand > 0);  // a trailing round has one candidate
        hc_candidate(r.x, r.y, r.theta, r.dt, r.dr, cand, &px, &py, &pth);
      }
    }
    double sn, cs;
    sincos(pth, &sn, &cs);
    if (tl == 0) {
      s_cur[half][0] = px;
      s_cur[half][1] = py;
      s_cur[half][2] = sn;
      s_cur[half][3] = cs;
      s_cur_go[half] = go ? 1 : 0;
    }
  }
  __syncthreads();

  const int t_entry = t;
  for (int k = 0;; ++k) {
    const int pk = k & 1;
    // the thread index as a value the compiler cannot see through: everything derived from it (beam addresses,
    // fingerprint multipliers, LDS offsets) is recomputed per super-step instead of being hoisted out of the loop and
    // held in registers across it -- hoisted, the kernel needs 237 VGPRs and a 1024-thread workgroup spills 85
    int t = t_entry;
    asm volatile("" : "+v"(t));
    const int lane = t & 63;
    const int tl = PAIR ? t - half * NTH : t;  // (this thread's place among its pose's threads, from the laundered index)
    // the kernel's arguments through a pointer the compiler cannot see through either: what the loop needs of them
    // is re-read from the kernarg segment (scalar loads, cached) where it is used instead of living in -- and being
    // spilled from -- scalar registers across the whole loop
    const __attribute__((address_space(4))) HcChainArgs *ap =
        (const __attribute__((address_space(4))) HcChainArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ap));
    // (the map view too: ~20 scalar registers that would otherwise live across the loop)
    MapViewCP mp = map_p;
    asm volatile("" : "+s"(mp));
    const MapView map = load_view(mp);
    if (stamp && k < 64) ap->stamps[8 * k + 0] = wall_clock64();
    // ---- the pose of super-step k: the entry the previous replay picked from the table made beside it
    const int sel = s_sel;
    if (sel == kSelStop) break;  // (uniform)
    int go, mode = 0;
    double px, py, sn, cs;
    if (sel >= 0) {
      const HcNextEntry &e = s_tab[sel];
      const unsigned flags = e.flags;
      const bool cert_end = CERT && s_cert_end != 0;  // (written with s_sel, before barrier (A))
      const int done = (int)((flags >> 3) & 1u) | (cert_end ? 1 : 0);
      go = (int)((flags >> (4 + half)) & 1u);
      px = e.p[half][0];
      py = e.p[half][1];
      sn = e.p[half][2];
      cs = e.p[half][3];
      if (wave == NT / 64 - 1 && lane < 1 + H) {
        // the new root (its bookkeeping half was written by the replay) and the pose, should the tree be re-scored:
        // two lanes of the LAST wave -- wave 0 has the scan's surplus beams and the replay, it is the one the
        // others wait for
        if (lane == 0) {
          HcState &w = s_st;
          const unsigned counts = e.counts;
          const unsigned failed = counts & 0xffffu;
          const double half = hc_pow_half(failed - w.failed);  // one exact halving per failed round (hc_round_of)
          w.x = e.x;
          w.y = e.y;
          w.theta = e.theta;
          w.dt = w.dt * half;
          w.dr = w.dr * half;
          w.recent_acc = 0.5 * w.recent_acc + (double)(counts >> 28);
          w.recent_n = 0.5 * w.recent_n + (double)((counts >> 16) & 0xfffu);
          w.failed = failed;
          w.shape = (int)(flags & 7u);
          w.done = done;
        } else {
          const int hh = lane - 1;  // (the pose of half hh, should the tree be re-scored)
          s_cur[hh][0] = e.p[hh][0];
          s_cur[hh][1] = e.p[hh][1];
          s_cur[hh][2] = e.p[hh][2];
          s_cur[hh][3] = e.p[hh][3];
          s_cur_go[hh] = (int)((flags >> (4 + hh)) & 1u);
        }
      }
      // ---- an inert root (hc_inert): what the reference still does from here is score this very pose
      // 6 x (limit - failed) + 1 more times, each a tie with the best score, each rejected.  Nobody scores them: the
      // bookkeeping workgroup writes those scorer calls into the observer's trace and adds them to the count
      // ... or a CERTIFIED root (see "certificate" below): the candidates differ from the best pose, but every beam of
      // every one of them ends in the best pose's cell -- the same terms, the same score, rejected all the same
      long long tail_calls = 0;
      if (done && ((flags & 64u) || cert_end)) tail_calls = 6ll * (long long)(ap->max_failed - (e.counts & 0xffffu)) + 1ll;
      if (done && init_slot && tail_calls > 0 && ap->trace) {  // (uniform: the whole workgroup is the bookkeeping one)
        __syncthreads();  // (s_st: the root state the tail starts from, completed just above by the last wave)
        HcTraceEntry *const trace = ap->trace + (size_t)blockIdx.y * (size_t)ap->trace_stride;
        const long long at0 = s_st.calls;
        const double best = s_st.best_prob, dt_e = s_st.dt, dr_e = s_st.dr;
        for (long long q = t; q < tail_calls; q += NT) {
          // scorer call q of the tail: candidate q % 6 of its round q / 6, whose steps were halved that often
          const double hlf = hc_pow_half((unsigned)(q / 6));
          HcTraceEntry te;
          hc_candidate(e.x, e.y, e.theta, dt_e * hlf, dr_e * hlf, (int)(q % 6), &te.x, &te.y, &te.theta);
          te.score = best;
          te.accepted = 0;
          te.pad = 0;
          if (at0 + q < ap->trace_cap) trace[at0 + q] = te;
          else host->error = 2;
        }
        __threadfence_system();
        __syncthreads();
      }
      if (done && init_slot && t == 0) {
        // ---- the chain is over: the last workgroup reports (every lane's trace stores first, then the result, then
        // the flag the host spins on)
        const HcState &w = s_st;  // (the bookkeeping half: this thread's own stores of the replay)
        __threadfence_system();
        HcHostOut *h = host;
        h->pose[0] = e.x;
        h->pose[1] = e.y;
        h->pose[2] = e.theta;
        h->best_prob = w.best_prob;
        h->calls = w.calls + tail_calls;
        h->tail_calls = tail_calls;
        h->evaluated = w.evaluated;
        h->steps = w.steps;
        h->rescored = w.rescored;
        h->gm_cx = -1;
        h->gm_cy = -1;
        h->gm_prob = -1.0;
        __hip_atomic_store(&h->done_seq, ap->epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (ap->n_done) {
          // a batch: the last chain to end tells the host.  This chain's result is on its way to the host BEFORE
          // it counts itself, so whoever sees the full count may announce everybody's
          __threadfence_system();
          const unsigned before = atomicAdd(ap->n_done, 1u);
          if (before + 1u == gridDim.y && ap->h_all_done)
            __hip_atomic_store(ap->h_all_done, ap->epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
      if (done) break;  // (uniform: every thread read the same entry)
    } else {
      if (sel == kSelRescore && init_slot) {
        // a re-scored tree: the bookkeeping workgroup scores its base pose once more (the rare super-step)
        if (wave == 0) {
          double sn_, cs_;
          sincos(s_st.theta, &sn_, &cs_);
          if (t == 0) {
            s_cur[0][0] = s_st.x;
            s_cur[0][1] = s_st.y;
            s_cur[0][2] = sn_;
            s_cur[0][3] = cs_;
            s_cur_go[0] = 1;
          }
        }
        __syncthreads();  // (uniform: the whole workgroup is the bookkeeping one)
      }
      go = s_cur_go[half];
      px = s_cur[half][0];
      py = s_cur[half][1];
      sn = s_cur[half][2];
      cs = s_cur[half][3];
      mode = sel == kSelRescore ? 1 : 0;
    }
    if (stamp && k < 64) ap->stamps[8 * k + 3] = wall_clock64();
    const unsigned tag = hc_tag(ap->tag_epoch, k);
    // (both barriers below are passed by every thread of the workgroup whether its pose is scored or not: the two
    // halves of a pair decide that independently)
    if (go) {
      // ---- score it: terms by beam, then the canonical sum (256 strided partials in ascending beam order, wave
      // butterfly, (g0+g1)+(g2+g3)) -- k_score_point's order; up to four beams per thread at a time, their cell
      // gathers issued together (hc_chain.hip)
      if (WIN) {
        // window OOPEs: k_score_window's per-beam value (occupancy_observation_probability.h:29-99), one beam at a
        // time -- a beam reads a window of cells, not one
        const double half_v = (ap->area[1] - ap->area[0]) / 2, half_h = (ap->area[3] - ap->area[2]) / 2;
        for (int b = tl; b < n; b += NTH) {
          double r_ = br, ca = bc, sa = bs, w = bw, f = bf;
          if (b != tl) {
            r_ = scan.range[b];
            ca = scan.cos_a[b];
            sa = scan.sin_a[b];
            w = scan.weight[b];
            f = scan.factor[b];
          }
          const double c = cs * ca - sn * sa;
          const double s = sn * ca + cs * sa;
          const double ox = px + r_ * c, oy = py + r_ * s;
          my_term[b] = window_probability<MODEL>(map, ap->oie, ap->oope, half_v, half_h, ox, oy) * w * f;
        }
      } else {
        // (five beams of 256 threads at a time: 1080 beams are then ONE round of gathers for every wave -- with
        // four, wave 0 went round twice for its 56 surplus beams while fifteen waves waited at the barrier)
        constexpr int UNR = NTH == 256 ? 5 : 4;
        for (int base = tl; base < n; base += UNR * NTH) {
          double4 cell[UNR];
          double w_[UNR], f_[UNR];
  #pragma unroll
          for (int j = 0; j < UNR; ++j) {
            const int b = base + j * NTH;
            w_[j] = 0.0;
            f_[j] = 0.0;
            cell[j] = make_double4(0.0, 0.0, 0.0, 0.0);
            if ((base - lane) + j * NTH >= n) continue;  // no lane of this wave has a beam in this slot
            const int bc_ = b < n ? b : n - 1;
            double r_ = br, ca = bc, sa = bs;
            w_[j] = bw;
            f_[j] = bf;
            if (j > 0 || base != tl) {
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
            const int b = base + j * NTH;
            if (b < n) my_term[b] = cell_probability<MODEL>(ap->oie, cell[j]) * w_[j] * f_[j];
          }
        }
      }
    }
    // ---- certificate (r06; the bookkeeping workgroup, idle behind the first super-step).  A match does not end when
    // it has converged but at its count of failed rounds; between the last acceptance (failed round ~10) and the inert
    // threshold (~50) lie rounds whose candidates differ from the best pose by less than any beam's distance from its
    // cell's edge: the same cells, the same terms, the same sum -- a tie with the best score, a rejection
    // (pose_enumeration_scan_matcher.h:58), in the canonical and in the beam-order sum alike.  A lone chain's tree of 42
    // round instances spends one super-step too many there (cfg2: 11.9 -> 10.5), the chains of a batch, with trees of a
    // few instances, twenty (K = 64: 56 -> 38).  So: for the ROOT pose of this super-step, how small do the steps
    // have to be for every candidate of a round based on it to end, beam by beam, in its own cells?  A beam's end point
    // moves by at most 1.01 dt (a translation candidate) or 1.5 r dr (a rotation candidate) plus rounding -- 2^-44 of
    // every magnitude involved, three orders above what the operations can lose; the distance from the nearest cell
    // edge is computed with the same slack against the true quotient's rounding.  t_t / t_r: the largest dt / dr all
    // beams allow (0: none).  They travel in this workgroup's granule; the replay (below) ends the chain when the walk
    // accepted nothing and the next root's steps are below them.
    double cert_t = 0.0, cert_r = 0.0;
    const bool cert_here = CERT && init_slot && half == 0 && sel >= 0 && !go && ap->inert_tail > 1;
    if (cert_here) {
      const double theta_abs = __builtin_fabs(s_tab[sel].theta);
      double tt = __builtin_inf(), tr = __builtin_inf();
      for (int b = tl; b < n; b += NTH) {
        double r_ = br, ca = bc, sa = bs;
        if (b != tl) {
          r_ = ldsc ? s_r[b] : scan.range[b];
          ca = ldsc ? s_ca[b] : scan.cos_a[b];
          sa = ldsc ? s_sa[b] : scan.sin_a[b];
        }
        double t1, t2;
        hc_cert_beam(px, py, sn, cs, theta_abs, r_, ca, sa, map.scale, map.inv_scale, &t1, &t2);
        tt = tt < t1 ? tt : t1;
        tr = tr < t2 ? tr : t2;
      }
      wave_min2(tt, tr);
      if (lane == 0) {
        s_cert[0][lwave] = tt;
        s_cert[1][lwave] = tr;
      }
    }
    __syncthreads();  // (B)
    if (cert_here && tl < 64) {
      double tt = tl < NTH / 64 ? s_cert[0][tl] : __builtin_inf(), tr = tl < NTH / 64 ? s_cert[1][tl] : __builtin_inf();
      wave_min2(tt, tr);
      cert_t = tt;
      cert_r = tr;
    }
    if (stamp && k < 64) ap->stamps[8 * k + 4] = wall_clock64();
    if (go) {
      if (SEQ) {
        // the reference's own order: one running sum over the beams (weighted_mean_point_probability_spe.h:108-124)
        if (tl == 0) {
          double acc = 0.0;
          int b = 0;
          for (; b + 8 <= n; b += 8) {
            const double t0 = my_term[b], t1 = my_term[b + 1], t2 = my_term[b + 2], t3 = my_term[b + 3];
            const double t4 = my_term[b + 4], t5 = my_term[b + 5], t6 = my_term[b + 6], t7 = my_term[b + 7];
            acc = acc + t0;
            acc = acc + t1;
            acc = acc + t2;
            acc = acc + t3;
            acc = acc + t4;
            acc = acc + t5;
            acc = acc + t6;
            acc = acc + t7;
          }
          for (; b < n; ++b) acc = acc + my_term[b];
          gran_store(&gran[pk * kGranRow + slot], (scan.tot_w == 0.0) ? __builtin_nan("") : acc / scan.tot_w, 0ull, tag);
        }
      } else if (tl < kSumLanes) {
        double acc = 0.0;
        unsigned long long h = 0ull;  // fingerprint of the term vector (score_device.h)
        unsigned k_lo = (2u * (unsigned)tl + 1u) * 0x9E3779B1u, k_hi = (2u * (unsigned)tl + 1u) * 0x85EBCA6Bu;
        for (int b = tl; b < n; b += kSumLanes) {
          const double term = my_term[b];
          acc = acc + term;
          if (verify) {
            h += term_fingerprint(term, k_lo, k_hi);
            k_lo += 2u * kSumLanes * 0x9E3779B1u;
            k_hi += 2u * kSumLanes * 0x85EBCA6Bu;
          }
        }
        wave_xor_sum_with(acc, h);
        if (lane == 0) {
          s_part[half][lwave] = acc;
          s_hpart[half][lwave] = h;
        }
      }
    }
    __syncthreads();  // (C) (the terms are rewritten by the next super-step)
    if (go) {
      if (!SEQ) {
        if (tl == 0) {
          const double total = (s_part[half][0] + s_part[half][1]) + (s_part[half][2] + s_part[half][3]);
          const unsigned long long fp =
              verify ? fold_fingerprint48(s_hpart[half][0] + s_hpart[half][1] + s_hpart[half][2] + s_hpart[half][3]) : 0ull;
          gran_store(&gran[pk * kGranRow + slot], (scan.tot_w == 0.0) ? __builtin_nan("") : total / scan.tot_w, fp, tag);
        }
        if (verify && mode && tl == 64) {
          // re-scored super-step: the reference's own order as well, one running sum over the beams
          // (read behind barrier (C): nobody writes the terms before the barrier that ends this super-step)
          double acc = 0.0;
          for (int b = 0; b < n; ++b) acc = acc + my_term[b];
          gran_store(&gseq[pk * kGranRow + slot], (scan.tot_w == 0.0) ? __builtin_nan("") : acc / scan.tot_w, 0ull, tag);
        }
      }
      if (stamp && k < 64) ap->stamps[8 * k + 5] = wall_clock64();
    } else if (tl == 0 && exists) {
      // nothing to score (behind the end of the chain, the surplus candidates of a trailing round, a smaller shape, the
      // bookkeeping workgroup behind the first super-step): the tag goes out all the same.  The sweepers wait for EVERY
      // workgroup of the grid in every super-step, so nobody -- the bookkeeping workgroup streaming an observer's
      // trace over PCIe least of all -- is ever more than one super-step behind the others, whose next-but-one
      // granules would overwrite what it still has to read
      // (the bookkeeping workgroup's granule carries its certificate: t_t where a score sits, the upper 48 bits of t_r
      // -- rounded down -- where a fingerprint does)
      gran_store(&gran[pk * kGranRow + slot], cert_here ? cert_t : 0.0,
                 cert_here ? (unsigned long long)__double_as_longlong(cert_r) >> 16 : 0ull, tag);
      if (verify && mode) gran_store(&gseq[pk * kGranRow + slot], 0.0, 0ull, tag);
    }

    if (wave != 0) {
      // ---- every other wave, while wave 0 waits for the scores: the table of next poses.  Lane = one (terminal
      // instance, outcome) pair of this super-step's tree (at most 7 x 42 = 294 of the 960 idle lanes of a lone chain)
      {
        const HcState &sp = s_st;
        const int shape = sp.shape;
        const int n_inst = (int)((ap->n_inst >> (8 * shape)) & 0xffull);
        const unsigned max_failed = ap->max_failed;
        for (int e = t - 64; e < 7 * n_inst; e += NT - 64) {
          const int ti = e / 7, out = e - 7 * ti;  // terminal round instance, its outcome
          HcInst in;
          {
            const unsigned long long *src = &ap->shapes[shape].inst[ti].w[0];
#pragma unroll
            for (int q = 0; q < 14; ++q) in.w[q] = src[q];
          }
          if (!(hc_is_root(in) || sp.failed + hc_nfail_parent(in) < max_failed)) continue;  // not reachable
          const bool trailing = hc_trailing(sp.failed + hc_nfail(in), max_failed);
          // only where a walk can END: a trailing round (one candidate: outcomes 0 and 1), or an outcome the shape
          // speculates no further on
          if (trailing ? out > 1 : hc_child(in, out) >= 0) continue;
          const HcRound r = hc_round_of(sp, in);
          const HcNextCore c = hc_next_core(sp, in, r, out, max_failed);
          HcNextEntry &w = s_tab[e];
          w.x = c.x;
          w.y = c.y;
          w.theta = c.theta;
          w.counts = hc_entry_counts(c, hc_nseg(in) + (out > 0 ? 1 : 0));
          // this workgroup's pose(s) in the tree hanging off that root (the bookkeeping workgroup scores nothing there)
          const bool inert = !c.done && ap->inert_tail && hc_inert(c.x, c.y, c.theta, c.dt, c.dr);
          unsigned flags = (unsigned)c.shape | (c.done || inert ? 8u : 0u) | (inert ? 64u : 0u);
#pragma unroll
          for (int hh = 0; hh < H; ++hh) {
            const int slot_h = PAIR ? 2 * (int)blockIdx.x + hh : slot;
            const int inst_h = slot_h / 6, cand_h = slot_h - 6 * inst_h;
            bool go = !c.done && !inert && !init_slot && slot_h < 6 * ap->max_inst;
            double px_ = c.x, py_ = c.y, pth_ = c.theta;
            if (go) {
              HcInst mine;
#pragma unroll
              for (int q = 0; q < 14; ++q) mine.w[q] = s_mine[hh][c.shape].w[q];
              go = inst_h < (int)((ap->n_inst >> (8 * c.shape)) & 0xffull) &&
                   (hc_is_root(mine) || c.failed + hc_nfail_parent(mine) < max_failed);  // else: behind the end of the chain
              if (go) {
                const HcRound r2 = hc_round_from(c.x, c.y, c.theta, c.dt, c.dr, c.failed, mine);
                go = !(hc_trailing(r2.failed, max_failed) && cand_h > 0);  // a trailing round has one candidate
                hc_candidate(r2.x, r2.y, r2.theta, r2.dt, r2.dr, cand_h, &px_, &py_, &pth_);
              }
            }
            double sn_ = 0.0, cs_ = 1.0;
            // (the bookkeeping workgroup scores nothing there: px_, py_, pth_ are the next ROOT's pose -- whose cells its
            // certificate is about, hence the sine and cosine)
            if (go || (CERT && init_slot && hh == 0 && ap->inert_tail > 1)) sincos(pth_, &sn_, &cs_);
            w.p[hh][0] = px_;
            w.p[hh][1] = py_;
            w.p[hh][2] = sn_;
            w.p[hh][3] = cs_;
            flags |= go ? (16u << hh) : 0u;
          }
          w.flags = flags;
        }
      }
    } else {
      // ---- wave 0: all scores of super-step k, then its replay
      const HcState &sp = s_st;  // (fields are read where they are used: a register copy of the struct is 34 VGPRs)
      const int n_inst = (int)((ap->n_inst >> (8 * sp.shape)) & 0xffull);
      // the round instances of this tree's shape, lane = instance: loads in flight while the scores arrive
      HcInst me;
      {
        const unsigned long long *src = &ap->shapes[sp.shape].inst[lane].w[0];
#pragma unroll
        for (int q = 0; q < 14; ++q) me.w[q] = src[q];
      }
      const bool active = lane < n_inst;
      const bool reach = active && (hc_is_root(me) || sp.failed + hc_nfail_parent(me) < ap->max_failed);
      const bool trailing = reach && hc_trailing(sp.failed + hc_nfail(me), ap->max_failed);
      const int bpi = hc_bp_inst(me);
      const int bp_slot = (!active || bpi < 0) ? -1 : 6 * bpi + hc_bp_cand(me);
      // ---- sweep: every workgroup of the grid (the bookkeeping one scored the initial pose / a re-scored base)
      const bool base_here = sp.first || sp.mode == 1;
      const int n_grid = 6 * ap->max_inst;  // scoring slots (+ the bookkeeping workgroup's)
      const bool rescored = !SEQ && verify && sp.mode == 1;
      bool failed = false;
      {
        const HcGranule *g0 = gran + pk * kGranRow;
        unsigned spins = 0;
        for (;;) {
          u32x4 g[G];
          const HcGranule *gp[G];
          bool ok = true;
#pragma unroll
          for (int q = 0; q < G; ++q) {
            const int i = lane + 64 * q;
            gp[q] = g0 + (i < n_grid ? i : kHcSlots - 1);  // (behind the grid: the bookkeeping workgroup's, once more)
          }
          gran_fetch(g, gp);  // (the loads and their wait in one statement: hc_resident_device.h)
#pragma unroll
          for (int q = 0; q < G; ++q) {
            const int i = lane + 64 * q;
            const int j = i < n_grid ? i : kHcSlots - 1;
            const bool here = gran_tag(g[q]) == tag;
            ok = ok && here;
            if (here) {
              s_sc[j] = gran_score(g[q]);
              s_hash[j] = gran_hash(g[q]);
            }
          }
          if (__all(ok)) break;
          ++spins;
          // (batches: the CUs are short of issue slots, not of polls -- a pause of ~0.2 us between two polls of a sweep: K = 8
          // 0.199 -> 0.193 ms per call, K = 16 0.283 -> 0.276; s_sleep 20: 0.198; a lone chain polls at once)
          if (BATCH) __builtin_amdgcn_s_sleep(8);
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
        // a workgroup of the grid is not on the chip (or the chain is longer than a tag can count): everybody
        // leaves, the host runs the match on the kernel chain
        if (lane == 0) {
          __hip_atomic_store(&rc->fail_epoch, ap->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&host->error, failed ? 4 : 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __threadfence_system();
          __hip_atomic_store(&host->done_seq, ap->epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
          s_sel = kSelStop;
        }
      } else {
        // ---- replay of super-step k's tree, lane = round instance (hc_chain.h)
        double root_prob = sp.first ? s_sc[kHcSlots - 1] : sp.best_prob;
        double s6[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) s6[c] = s_sc[6 * lane + c];
        const double enter = bp_slot < 0 ? root_prob : s_sc[bp_slot];  // canonical (reported) score entering the round
        double run = enter;
        int out = 0;
        unsigned accmask = 0u;
        bool ambiguous = false;
        unsigned long long run_hash = 0ull;
        if (!SEQ && verify) {
          // slot kHcSlots-1 holds the base pose of a re-scored tree (or the initial pose): its sums head the path
          const unsigned long long root_hash = base_here ? s_hash[kHcSlots - 1] : sp.best_hash;
          unsigned long long hb = bp_slot < 0 ? root_hash : s_hash[bp_slot];
          unsigned long long h6[6];
#pragma unroll
          for (int c = 0; c < 6; ++c) h6[c] = s_hash[6 * lane + c];
          if (!rescored) {
            // decisions from the canonical sums; a comparison closer than the two summation orders can differ
            // (2^-40, relative) between poses whose term vectors differ is one the tree sum cannot settle
            int amb = 0;
#pragma unroll
            for (int c = 0; c < 6; ++c) {
              const double s = s6[c];
              const double diff = __builtin_fabs(s - run);
              const double as = __builtin_fabs(s), ab = __builtin_fabs(run);
              const int live = (c == 0) | (int)!trailing;
              const int close = (int)(diff <= (as > ab ? as : ab) * 9.094947017729282e-13);  // NaN: false, a rejection
              // (equal fingerprints with different sums: not identical vectors -- a collision, equally unsettled)
              // (bitwise, not short-circuit: one straight line of compares instead of six nested exec-mask branches)
              amb |= live & close & ((int)(h6[c] != hb) | (int)(__double_as_longlong(s) != __double_as_longlong(run)));
              const bool acc = (live & (int)(run < s)) != 0;  // strict: ties are rejections (pose_enumeration_scan_matcher.h:58)
              run = acc ? s : run;
              hb = acc ? h6[c] : hb;
              out = acc ? c + 1 : out;
              accmask |= acc ? 1u << c : 0u;
            }
            ambiguous = amb != 0;
          } else {
            // re-scored tree: the same comparisons on the beam-order sums.  Their granules were stored next to the
            // canonical ones by other lanes: wait for their tags as well, one granule at a time (the rare step: a
            // rolled loop that costs no registers)
            const HcGranule *q0 = gseq + pk * kGranRow;
            double bdec = 0.0;
#pragma unroll 1
            for (int c = -1; c < 6; ++c) {
              const int j = c < 0 ? (bp_slot < 0 ? kHcSlots - 1 : bp_slot) : (active ? 6 * lane + c : kHcSlots - 1);
              double sd = 0.0;
              for (unsigned spins = 0;; ++spins) {
                u32x4 g[1];
                const HcGranule *gp[1] = {q0 + j};
                gran_fetch(g, gp);
                sd = gran_score(g[0]);
                if (__all(gran_tag(g[0]) == tag) || spins > ap->spin_limit) break;  // (cannot run out: the canonical granules of
              }                                                                 // the same workgroups are here already)
              if (c < 0) {
                bdec = sd;
                continue;
              }
              const bool acc = active && (c == 0 || !trailing) && bdec < sd;
              bdec = acc ? sd : bdec;
              run = acc ? s_sc[6 * lane + c] : run;  // the canonical sum of the pose accepted last: stored and reported
              hb = acc ? s_hash[6 * lane + c] : hb;
              out = acc ? c + 1 : out;
              accmask |= acc ? 1u << c : 0u;
            }
          }
          run_hash = hb;
        } else {
#pragma unroll
          for (int c = 0; c < 6; ++c)
            if ((c == 0 || !trailing) && run < s6[c]) {  // strict: ties are rejections
              run = s6[c];
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
        // exactly one lane is terminal: the walk's last round
        const int tl = tmask ? __ffsll((long long)tmask) - 1 : 0;
        if (stamp && k < 64) ap->stamps[8 * k + 7] = wall_clock64();
        // checked default mode: a comparison on the walked path that the tree sum cannot settle -> the same tree is
        // scored once more, in beam order as well, and decided from those sums
        const bool dirty = !SEQ && verify && sp.mode == 0 && __ballot(valid && ambiguous) != 0ull;
        // the terminal lane's outcome and best score in every lane; where the next tree hangs is in the table
        const int out_t = bcast_i(out, tl);
        const double run_t = bcast(run, tl);
        const int depth_t = bcast_i(hc_depth(me), tl);
        const bool trailing_t = bcast_i(trailing ? 1 : 0, tl) != 0;
        const unsigned long long run_hash_t = (unsigned long long)bcast_ll((long long)run_hash, tl);
        if (init_slot && ap->trace && !dirty) {
          // ---- the last workgroup keeps the books (it scores nothing after the first super-step)
          HcTraceEntry *const trace = ap->trace + (size_t)blockIdx.y * (size_t)ap->trace_stride;
          const long long base = sp.calls + (sp.first ? 1 : 0);
          if (sp.first && lane == 0 && ap->trace_cap > 0) {
            HcTraceEntry e{sp.x, sp.y, sp.theta, root_prob, 1, 0};
            trace[0] = e;
          }
          if (valid) {
            const HcRound rr = hc_round_of(sp, me);
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
        // ---- the certificate's verdict (every workgroup reads the same granule, every replay decides alike): the walk
        // accepted nothing (it ends on the root's own line of failed rounds: the next root is the same pose with smaller
        // steps), and those steps are below what the bookkeeping workgroup certified for this pose -- the chain is over
        bool cert_end = false;
        if (CERT && ap->inert_tail > 1 && !sp.first && sp.mode == 0 && !dirty && tmask != 0ull) {
          const int nseg_t = bcast_i(hc_nseg(me), tl), nfail_t = bcast_i((int)hc_nfail(me), tl);
          const unsigned failed_next = sp.failed + (unsigned)nfail_t + 1u;
          if (nseg_t == 0 && out_t == 0 && !trailing_t && failed_next < ap->max_failed) {
            const double hlf = hc_pow_half((unsigned)nfail_t + 1u);
            const double t_t = s_sc[kHcSlots - 1];
            const double t_r = __longlong_as_double((long long)(s_hash[kHcSlots - 1] << 16));
            cert_end = sp.dt * hlf < t_t && sp.dr * hlf < t_r;
          }
        }
        if (stamp && k < 64) ap->stamps[8 * k + 2] = wall_clock64();
        if (lane == 0) {
          s_cert_end = cert_end ? 1 : 0;
          // the bookkeeping half of the next root state (`sp` above is this very object: the trace was written from
          // the old state first); the other half -- pose, steps, shape -- is copied from the table behind the barrier
          HcState &w = s_st;
          if (tmask == 0ull) {  // cannot happen (the root round is always on the path): stop instead of looping
            if (init_slot) {
              __hip_atomic_store(&host->error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
              __threadfence_system();
              __hip_atomic_store(&host->done_seq, ap->epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            s_sel = kSelStop;
          } else if (!dirty) {
            w.best_prob = run_t;
            w.calls = sp.calls + (sp.first ? 1 : 0) + 6ll * depth_t + (trailing_t ? 1 : 6);
            w.evaluated = sp.evaluated + 6ll * n_inst + (sp.first ? 1 : 0);
            w.first = 0;
            w.mode = 0;
            w.steps = sp.steps + 1;
            if (!SEQ && verify) w.best_hash = run_hash_t;
            s_sel = 7 * tl + out_t;
          } else {
            // same root, same tree, same `first`
            w.mode = 1;
            w.steps = sp.steps + 1;
            w.evaluated = sp.evaluated + 6ll * n_inst + 1;
            w.rescored = sp.rescored + 1;
            s_sel = kSelRescore;
          }
        }
      }
    }
    __syncthreads();  // (A) the table is complete and the replay has picked its entry
  }
}

// dynamic LDS of a workgroup: the beams' terms, with lds_consts range, cosine and sine of the beams behind every
// thread's first one, and the table of next poses (7 entries per round instance of the largest shape)
size_t hc_resident_tab_offset(int nt, int n_beams, bool lds_consts, bool pair) {  // (in doubles)
  const size_t n = (size_t)(n_beams > 0 ? n_beams : 1);
  const size_t nth = (size_t)(pair ? nt / 2 : nt);  // threads per pose
  const size_t more = lds_consts && n > nth ? n - nth : 0;
  return (pair ? 2 : 1) * n + 3 * more;
}
size_t hc_resident_lds_bytes(int nt, int n_beams, bool lds_consts, int max_inst, bool pair) {
  return sizeof(double) * hc_resident_tab_offset(nt, n_beams, lds_consts, pair) +
         (pair ? sizeof(HcNextEntryT<2>) : sizeof(HcNextEntryT<1>)) * 7 * (size_t)max_inst;
}

#define HCR_LAUNCH(NTV, GV)                                                                                     \
  do {                                                                                                          \
    if (e0 || e1)                                                                                               \
      hipExtLaunchKernelGGL((k_hc_chain_resident<MODEL, NTV, SEQ, BATCH, GV>), dim3(grid, n_chains), dim3(NTV), shm, stream, e0, e1, 0, a); \
    else                                                                                                        \
      hipLaunchKernelGGL((k_hc_chain_resident<MODEL, NTV, SEQ, BATCH, GV>), dim3(grid, n_chains), dim3(NTV), shm, stream, a);     \
  } while (0)

#define HCR_LAUNCH_PAIR(GV)                                                                                     \
  do {                                                                                                          \
    if (e0 || e1)                                                                                               \
      hipExtLaunchKernelGGL((k_hc_chain_resident<MODEL, 512, false, BATCH, GV, false, BATCH>), dim3(grid, n_chains), dim3(512), shm, stream, e0, e1, 0, a); \
    else                                                                                                        \
      hipLaunchKernelGGL((k_hc_chain_resident<MODEL, 512, false, BATCH, GV, false, BATCH>), dim3(grid, n_chains), dim3(512), shm, stream, a);     \
  } while (0)

// granules per sweeping lane for a grid of `grid` workgroups
static int gran_per_lane(int grid) { return grid <= 128 ? 2 : (grid <= 256 ? 4 : 7); }

#define HCR_LAUNCH_WIN(NTV)                                                                                      \
  do {                                                                                                          \
    if (e0 || e1)                                                                                               \
      hipExtLaunchKernelGGL((k_hc_chain_resident<MODEL, NTV, false, false, 4, true>), dim3(grid, n_chains), dim3(NTV), shm, stream, e0, e1, 0, a); \
    else                                                                                                        \
      hipLaunchKernelGGL((k_hc_chain_resident<MODEL, NTV, false, false, 4, true>), dim3(grid, n_chains), dim3(NTV), shm, stream, a);     \
  } while (0)

// the window OOPEs: lone chains of at most 256 workgroups, default sum order
template <int MODEL>
static hipError_t launch_res_win(const HcChainArgs &a_in, int nt, hipStream_t stream, hipEvent_t e0, hipEvent_t e1,
                                 int n_chains) {
  HcChainArgs a = a_in;
  const int grid = 6 * a.max_inst + 1;
  if (grid > 256) return hipErrorInvalidValue;
  const size_t shm = hc_resident_lds_bytes(nt, a.scan.n, false, a.max_inst, false);  // (the window form keeps no beam constants in LDS)
  a.tab_offset = (int)hc_resident_tab_offset(nt, a.scan.n, false, false);
  if (nt == 1024) HCR_LAUNCH_WIN(1024);
  else if (nt == 256) HCR_LAUNCH_WIN(256);
  else HCR_LAUNCH_WIN(512);
  return hipGetLastError();
}
#undef HCR_LAUNCH_WIN

template <int MODEL, bool SEQ, bool BATCH>
static hipError_t launch_res(const HcChainArgs &a_in, int nt, hipStream_t stream, hipEvent_t e0, hipEvent_t e1,
                             int n_chains) {
  HcChainArgs a = a_in;
  const bool pair = BATCH && a.pair != 0;
  const int slots = 6 * a.max_inst + 1;
  const int grid = pair ? 3 * a.max_inst + 1 : slots;  // (a pair: two scoring slots per workgroup + the bookkeeping one)
  const size_t shm = hc_resident_lds_bytes(nt, a.scan.n, a.lds_consts != 0, a.max_inst, pair);
  a.tab_offset = (int)hc_resident_tab_offset(nt, a.scan.n, a.lds_consts != 0, pair);
  const int g = gran_per_lane(slots);
  if (pair) {
    if (nt != 512) return hipErrorInvalidValue;
    if (g == 2) HCR_LAUNCH_PAIR(2);
    else if (g == 4) HCR_LAUNCH_PAIR(4);
    else HCR_LAUNCH_PAIR(7);
    return hipGetLastError();
  }
  // (workgroup sizes and sweep widths that go together: a lone chain is 253 x 1024 threads, a batch's chains are
  // narrower trees of narrower workgroups)
  if (nt == 1024) {
    if (g == 2) HCR_LAUNCH(1024, 2);
    else if (g == 4) HCR_LAUNCH(1024, 4);
    else return hipErrorInvalidValue;  // 385 workgroups of 1024 threads are not resident together
  } else if (nt == 256) {
    if (g == 2) HCR_LAUNCH(256, 2);
    else if (g == 4) HCR_LAUNCH(256, 4);
    else HCR_LAUNCH(256, 7);
  } else {
    if (g == 2) HCR_LAUNCH(512, 2);
    else if (g == 4) HCR_LAUNCH(512, 4);
    else HCR_LAUNCH(512, 7);
  }
  return hipGetLastError();
}
#undef HCR_LAUNCH
#undef HCR_LAUNCH_PAIR

hipError_t launch_hc_chain_resident(const HcChainArgs &a, int cell_model, int nt, hipStream_t stream, hipEvent_t e0,
                                    hipEvent_t e1, int n_chains) {
  if (!a.rctl) return hipErrorInvalidValue;
  if (a.oope != SLAMHIP_OOPE_OBSTACLE) {
    if (a.jobs || a.seq) return hipErrorInvalidValue;
    if (cell_model == SLAMHIP_CELL_OCC) return launch_res_win<SLAMHIP_CELL_OCC>(a, nt, stream, e0, e1, n_chains);
    if (cell_model == SLAMHIP_CELL_TBM) return launch_res_win<SLAMHIP_CELL_TBM>(a, nt, stream, e0, e1, n_chains);
    return hipErrorInvalidValue;
  }
  if (a.jobs) {
    if (a.seq) return hipErrorInvalidValue;
    if (cell_model == SLAMHIP_CELL_OCC) return launch_res<SLAMHIP_CELL_OCC, false, true>(a, nt, stream, e0, e1, n_chains);
    if (cell_model == SLAMHIP_CELL_TBM) return launch_res<SLAMHIP_CELL_TBM, false, true>(a, nt, stream, e0, e1, n_chains);
    return hipErrorInvalidValue;
  }
  if (cell_model == SLAMHIP_CELL_OCC)
    return a.seq ? launch_res<SLAMHIP_CELL_OCC, true, false>(a, nt, stream, e0, e1, n_chains)
                 : launch_res<SLAMHIP_CELL_OCC, false, false>(a, nt, stream, e0, e1, n_chains);
  if (cell_model == SLAMHIP_CELL_TBM)
    return a.seq ? launch_res<SLAMHIP_CELL_TBM, true, false>(a, nt, stream, e0, e1, n_chains)
                 : launch_res<SLAMHIP_CELL_TBM, false, false>(a, nt, stream, e0, e1, n_chains);
  return hipErrorInvalidValue;
}

// the instantiation launch_res / launch_res_win pick for a workgroup size and a sweep width
template <int M, bool B>
static const void *res_fn(int nt, int g) {
  if (nt == 1024)
    return g == 2 ? (const void *)k_hc_chain_resident<M, 1024, false, B, 2>
                  : (g == 4 ? (const void *)k_hc_chain_resident<M, 1024, false, B, 4> : nullptr);
  if (nt == 256)
    return g == 2 ? (const void *)k_hc_chain_resident<M, 256, false, B, 2>
                  : (g == 4 ? (const void *)k_hc_chain_resident<M, 256, false, B, 4>
                            : (const void *)k_hc_chain_resident<M, 256, false, B, 7>);
  return g == 2 ? (const void *)k_hc_chain_resident<M, 512, false, B, 2>
                : (g == 4 ? (const void *)k_hc_chain_resident<M, 512, false, B, 4>
                          : (const void *)k_hc_chain_resident<M, 512, false, B, 7>);
}
template <int M>
static const void *res_fn_pair(int g) {
  return g == 2 ? (const void *)k_hc_chain_resident<M, 512, false, true, 2, false, true>
                : (g == 4 ? (const void *)k_hc_chain_resident<M, 512, false, true, 4, false, true>
                          : (const void *)k_hc_chain_resident<M, 512, false, true, 7, false, true>);
}
template <int M>
static const void *res_fn_win(int nt) {
  return nt == 1024 ? (const void *)k_hc_chain_resident<M, 1024, false, false, 4, true>
                    : (nt == 256 ? (const void *)k_hc_chain_resident<M, 256, false, false, 4, true>
                                 : (const void *)k_hc_chain_resident<M, 512, false, false, 4, true>);
}

// Workgroups of `nt` threads (scoring `n_beams`, trees of `max_inst` round instances: hc_resident_lds_bytes of dynamic
// LDS) the device keeps resident at once: the occupancy query of the instantiation that will be LAUNCHED, the
// register-file rule of MI355X_MICROARCH.md ("Residency and cooperative launch": the API can be one block per CU high
// above 80 SGPRs; this kernel's waves also need <= 128 VGPRs at 1024 threads), minus ONE CU's worth of workgroups of
// margin -- a grid that needs every slot of the chip waits for any other kernel's last workgroup to leave.
hipError_t hc_resident_capacity(int cell_model, int nt, bool batch, bool window, int n_beams, bool lds_consts, int max_inst,
                                int *out_wgs, int *out_per_cu, bool pair) {
  const size_t lds_bytes = hc_resident_lds_bytes(nt, n_beams, lds_consts, max_inst, pair);
  int dev = 0, cus = 0, per_cu = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (e != hipSuccess) return e;
  const void *fn = nullptr;
  const int g = window ? 4 : gran_per_lane(6 * max_inst + 1);
  if (window) fn = cell_model == SLAMHIP_CELL_TBM ? res_fn_win<SLAMHIP_CELL_TBM>(nt) : res_fn_win<SLAMHIP_CELL_OCC>(nt);
  else if (batch && pair) fn = cell_model == SLAMHIP_CELL_TBM ? res_fn_pair<SLAMHIP_CELL_TBM>(g) : res_fn_pair<SLAMHIP_CELL_OCC>(g);
  else if (batch) fn = cell_model == SLAMHIP_CELL_TBM ? res_fn<SLAMHIP_CELL_TBM, true>(nt, g) : res_fn<SLAMHIP_CELL_OCC, true>(nt, g);
  else fn = cell_model == SLAMHIP_CELL_TBM ? res_fn<SLAMHIP_CELL_TBM, false>(nt, g) : res_fn<SLAMHIP_CELL_OCC, false>(nt, g);
  if (!fn) return hipErrorInvalidValue;
  e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, nt, lds_bytes);
  if (e != hipSuccess) return e;
  const int by_waves = 2048 / nt;  // 128-VGPR waves: four per SIMD
  per_cu = per_cu < by_waves ? per_cu : by_waves;
  if (per_cu > 6) per_cu = 6;      // floor(800 / (ceil(sgpr / 16) * 16 + 16)) at ~106 SGPRs
  *out_wgs = per_cu * (cus - 1);
  if (out_per_cu) *out_per_cu = per_cu;
  return hipSuccess;
}

}  // namespace slamhip
