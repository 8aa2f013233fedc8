// hc_chain.h -- the hill-climbing accept chain kept ON the device.
//
// What it restates (paths relative to the reference root):
//   PoseEnumerationScanMatcher::process_scan      src/core/scan_matchers/pose_enumeration_scan_matcher.h:31-77
//   Distorsion1DPoseEnumerator +
//   FailedRoundsLimitedPoseEnumerator (HC)        src/core/scan_matchers/hill_climbing_scan_matcher.h:10-126
//
// Why.  The host-driven matcher (matchers.h) needs ~15 launches per process_scan and pays a host round
// trip for each (launch, completion flag over PCIe, replay, re-plan: ~16 us against ~8 us of kernel).
// Here one process_scan is a CHAIN OF KERNELS on one stream with no host in between: kernel k scores the
// speculation tree of super-step k; the prologue of kernel k+1 (one wave, repeated by every workgroup so
// that nothing has to be broadcast) replays the accept chain over those scores, derives the enumerator
// state the walk ended in and the candidate this workgroup scores next.  A dependent kernel boundary
// costs ~1.5 us and gives full visibility of the previous kernel's stores, cheaper than any in-kernel
// grid barrier on this part (MI355X_MICROARCH.md: boundary 1.45 us, barrier-xcd 4.1-4.7 us).
//
// The speculation tree of a super-step is an instance of a static SHAPE: a set of round instances
// (parent, outcome of the parent that leads here) chosen best-first by outcome probability for one
// per-candidate acceptance rate, exactly like SpecTree::build_rounds chooses them on the host -- but
// without poses, so it is built once per matcher for a few acceptance-rate buckets and lives in HBM.
// A hill-climbing round has six candidates that depend only on the state at the round's start (base
// pose latched once per round, Q4) and seven outcomes (0 = all six rejected: both steps halve and the
// failed-round counter goes up at the start of the next round; j = candidate j-1 accepted last: it is
// the next base).  The state reached by any path is therefore a closed form of the root state and the
// path's accepted moves -- the functions below -- and whether a round instance lies on the real path
// follows from the scores alone: instance i enters with the score of the candidate its path accepted
// last, its own outcome is a running strict maximum over its six scores, and it is on the path when
// every ancestor produced the outcome the path prescribes (seven wave ballots against static masks).
//
// Everything here is `__host__ __device__` so that tests/native/hc_chain_test.cpp can run the same
// code on the CPU against HillClimbingPoseEnumerator (matchers.h).
#pragma once

#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HC_HD __host__ __device__ __forceinline__
#else
#define HC_HD inline
#endif

namespace slamhip {

constexpr int kHcMaxInst = 64;    // round instances per shape: one lane of a wave64 each
constexpr int kHcDefaultInst = 42; // instances per shape by default: 6 x 42 + 1 = 253 workgroups, one per CU of an MI355X
constexpr int kHcMaxSeg = 14;     // accepted moves on one path (deeper children are not speculated)
constexpr int kHcShapes = 8;      // acceptance-rate buckets
constexpr int kHcSlots = 6 * kHcMaxInst + 1;  // workgroups per super-step; the last one scores the initial pose

// per-candidate acceptance rate of bucket b: 0.5 / 2^(7-b)  (0.0039 .. 0.5, the host matcher's clamp range)
HC_HD double hc_bucket_rate(int b) { return 0.5 / (double)(1 << (kHcShapes - 1 - b)); }
// nearest bucket in the log domain (boundaries at rate(b) * sqrt(2)) of the rate num / den, den > 0: seven
// independent comparisons by cross-multiplication -- no division, no loop-carried dependence
HC_HD int hc_bucket_of_ratio(double num, double den) {
  int b = 0;
  for (int j = 0; j < kHcShapes - 1; ++j) b += (num > hc_bucket_rate(j) * 1.4142135623730951 * den) ? 1 : 0;
  return b;
}
HC_HD int hc_bucket_of(double p) { return hc_bucket_of_ratio(p, 1.0); }

// one round instance of a shape: 16 eight-byte words, so that a lane pulls its record out of LDS with
// sixteen loads in flight and picks fields with shifts -- field-by-field LDS reads of a byte-packed struct
// (each waited for before the next) made the replay 2 us long.
//   w[0..6]  need[o]: ancestors (bit = instance) that must have produced outcome o
//   w[7]     child[0..3], w[8] child[4..6], bp_inst : int16 each; child = instance reached through outcome
//            o, -1 = not speculated; bp_inst = instance whose accepted candidate supplies the entering best
//            score, -1 = the root's
//   w[9]     bytes: bp_cand, nfail (all-rejected rounds on the path root -> this instance), nfail_parent,
//            depth (rounds on the path before this one), nseg (accepted moves on the path), is_root,
//            tail_fail (all-rejected rounds after the last accepted move), parent (255 = none)
//   w[10,11] seg_fail[k]: all-rejected rounds between the previous accepted move and move k (bytes)
//   w[12,13] seg_out[k]: outcome 1..6 of move k (bytes)
struct HcInst {
  unsigned long long w[16];
};
static_assert(sizeof(HcInst) == 128, "HcInst layout");

HC_HD int hc_child(const HcInst &i, int out) {
  const unsigned long long v = out < 4 ? i.w[7] : i.w[8];
  return (int)(short)(unsigned short)(v >> (16 * (out & 3)));
}
HC_HD int hc_bp_inst(const HcInst &i) { return (int)(short)(unsigned short)(i.w[8] >> 48); }
HC_HD int hc_byte9(const HcInst &i, int k) { return (int)((i.w[9] >> (8 * k)) & 0xffull); }
HC_HD int hc_bp_cand(const HcInst &i) { return hc_byte9(i, 0); }
HC_HD unsigned hc_nfail(const HcInst &i) { return (unsigned)hc_byte9(i, 1); }
HC_HD unsigned hc_nfail_parent(const HcInst &i) { return (unsigned)hc_byte9(i, 2); }
HC_HD int hc_depth(const HcInst &i) { return hc_byte9(i, 3); }
HC_HD int hc_nseg(const HcInst &i) { return hc_byte9(i, 4); }
HC_HD bool hc_is_root(const HcInst &i) { return hc_byte9(i, 5) != 0; }
HC_HD unsigned hc_tail_fail(const HcInst &i) { return (unsigned)hc_byte9(i, 6); }
HC_HD int hc_parent(const HcInst &i) { return hc_byte9(i, 7) == 255 ? -1 : hc_byte9(i, 7); }
HC_HD unsigned hc_seg_fail(const HcInst &i, int k) {
  return (unsigned)(((k < 8 ? i.w[10] : i.w[11]) >> (8 * (k & 7))) & 0xffull);
}
HC_HD int hc_seg_out(const HcInst &i, int k) {
  return (int)(((k < 8 ? i.w[12] : i.w[13]) >> (8 * (k & 7))) & 0xffull);
}
// host-side setters (shape builder)
inline void hc_set16(unsigned long long &w, int slot, int v) {
  w = (w & ~(0xffffull << (16 * slot))) | ((unsigned long long)(unsigned short)(short)v << (16 * slot));
}
inline void hc_set8(unsigned long long &w, int slot, unsigned v) {
  w = (w & ~(0xffull << (8 * slot))) | ((unsigned long long)(v & 0xff) << (8 * slot));
}
inline void hc_set_child(HcInst &i, int out, int v) { hc_set16(out < 4 ? i.w[7] : i.w[8], out & 3, v); }
inline void hc_set_bp_inst(HcInst &i, int v) { hc_set16(i.w[8], 3, v); }
inline void hc_set_byte9(HcInst &i, int k, unsigned v) { hc_set8(i.w[9], k, v); }
inline void hc_set_seg(HcInst &i, int k, unsigned fail, unsigned out) {
  hc_set8(k < 8 ? i.w[10] : i.w[11], k & 7, fail);
  hc_set8(k < 8 ? i.w[12] : i.w[13], k & 7, out);
}

struct HcShape {
  int n_inst;
  int pad[31];
  HcInst inst[kHcMaxInst];
};

// enumerator + matcher state at a round boundary, "post-bump": failed/dt/dr are what the round about to
// start will use (the reference bumps the counter and halves the steps inside the next() that opens the
// round, hill_climbing_scan_matcher.h:89-101)
struct HcState {
  double x, y, theta;      // best pose = base of the round about to start
  double best_prob;
  double dt, dr;
  double recent_acc, recent_n;  // acceptance-rate estimate (MatchJob::consume)
  long long calls;         // scorer calls so far (= trace position)
  long long evaluated;     // poses scored on the GPU so far
  unsigned failed;         // failed rounds counted so far
  int shape;               // shape of the tree this state is the root of
  int done;                // the enumerator has no next candidate
  int first;               // the initial pose has not been scored yet (it rides in slot kHcSlots-1)
  int steps;
  int mode;                // 1: this super-step's tree is scored in beam order too and decided from those sums
                           // (the previous replay met an ambiguous comparison); slot kHcSlots-1 then re-scores the base
  // GMapping OOPE only: the cache entry the last replayed pose left (gmapping_occupancy_observation_pe.h:43-44)
  int carry_cx, carry_cy;
  double carry_prob;
  unsigned long long best_hash;  // term-vector hash of the best pose (checked default mode)
  long long rescored;            // super-steps scored twice so far
};
static_assert(sizeof(HcState) == 136, "HcState layout");

// candidate c (0..5) of a round with base (x, y, theta) and steps dt, dr: +X -Y +Th -X +Y -Th
// (action id % 3 = axis, id % 2 = sign; frame rotation always 0, Q5), in the reference's operation
// order (hill_climbing_scan_matcher.h:39-60)
HC_HD void hc_candidate(double x, double y, double theta, double dt, double dr, int c, double *ox, double *oy,
                        double *ot) {
  const double dir = (c % 2) ? -1.0 : 1.0;
  const double fcos = 1.0, fsin = 0.0;  // std::cos(0.0), std::sin(0.0)
  const int axis = c % 3;
  // the three cases of the reference's switch, evaluated side by side and selected (no divergent branch)
  const double x0 = x + fcos * dir * dt, y0 = y + fsin * dir * dt;
  const double x1 = x + -fsin * dir * dt, y1 = y + fcos * dir * dt;
  const double t2 = theta + dir * dr;
  *ox = axis == 0 ? x0 : (axis == 1 ? x1 : x);
  *oy = axis == 0 ? y0 : (axis == 1 ? y1 : y);
  *ot = axis == 2 ? t2 : theta;
}

// A round whose six candidates ARE its base pose, bit for bit: the steps have been halved below half an ulp of every
// coordinate (0.1 x 2^-k against poses of metres and radians: k ~ 47 ... 57).  The reference scores each of them all
// the same -- the same pose, so the same score as the best one's, a tie, a rejection (pose_enumeration_scan_matcher.h:58)
// -- and every later round is the same round with smaller steps: from here on the match is 6 x (limit - failed) + 1
// scorer calls that return the best score, and then the result.  The co-resident chain ends there in closed form
// (hc_resident.hip, HcChainArgs::inert_tail; long before that it ends on a CERTIFIED root, see there), and so does the
// chain of kernels (hc_chain.hip); the host-driven form scores the tail -- same traces.
HC_HD bool hc_inert(double x, double y, double theta, double dt, double dr) {
  union U {
    double d;
    unsigned long long u;
  };
  U bx, by, bt;
  bx.d = x;
  by.d = y;
  bt.d = theta;
  bool same = true;
  for (int c = 0; c < 6; ++c) {
    U cx, cy, ct;
    hc_candidate(x, y, theta, dt, dr, c, &cx.d, &cy.d, &ct.d);
    same = same && cx.u == bx.u && cy.u == by.u && ct.u == bt.u;
  }
  return same;
}

// The CERTIFICATE of a pose (hc_resident.hip): the largest steps for which no candidate of a round based on the pose
// (px, py, theta; sn / cs its sine and cosine) can move THIS beam's end point out of the cell it ends in -- so that the
// 1-cell scorer's term, a function of the cell alone, cannot change.  A translation candidate moves the end point by at
// most 1.01 dt, a rotation candidate by at most 1.5 r dr (sqrt(2) r dr, the error of two sincos, roundings), plus
// rounding bounded by 2^-44 of every magnitude involved -- r (1 + |theta|), the end point, the pose: ~500 x what the
// operations can lose --; that has to stay below the end point's distance from the nearest edge of its cell, computed
// with the same slack against the rounding of the true quotient world_to_cell takes the floor of.  *t_t / *t_r: the
// beam's bounds on dt / dr (0: none -- the end point sits on an edge, or something is not finite).  The minimum over the
// beams certifies the pose.  One definition for the kernel and for tests/native/hc_chain_test.cpp, which holds it
// against the plain accept loop.
HC_HD void hc_cert_beam(double px, double py, double sn, double cs, double theta_abs, double r_, double ca, double sa,
                        double scale, double inv_scale, double *t_t, double *t_r) {
  const double c = cs * ca - sn * sa;
  const double s_ = sn * ca + cs * sa;
  const double wx = px + r_ * c, wy = py + r_ * s_;
  const double qx = wx * inv_scale, qy = wy * inv_scale;
  const double fx = qx - __builtin_floor(qx), fy = qy - __builtin_floor(qy);
  const double mx = fx < 1.0 - fx ? fx : 1.0 - fx, my = fy < 1.0 - fy ? fy : 1.0 - fy;
  const double edge = (mx < my ? mx : my) * scale;
  const double ar = __builtin_fabs(r_);
  const double slack = (ar * (1.0 + theta_abs) + __builtin_fabs(wx) + __builtin_fabs(wy) + __builtin_fabs(px) +
                        __builtin_fabs(py) + 1.0) * 0x1p-44;
  const double avail = edge - slack;
  const bool ok = avail > 0.0;  // (false for a NaN)
  *t_t = ok ? avail * 0.99 : 0.0;
  *t_r = ok ? (ar > 0.0 ? avail / (1.5 * ar) : __builtin_inf()) : 0.0;
}

// 2^-k as a double (k < 1000): halving a normal double k times is one exact multiplication by it
HC_HD double hc_pow_half(unsigned k) {
  union {
    unsigned long long u;
    double d;
  } v;
  v.u = (unsigned long long)(1023 - k) << 52;
  return v.d;
}

// state at the start of round instance `in`, from the root state
struct HcRound {
  double x, y, theta, dt, dr;
  unsigned failed;
};
HC_HD HcRound hc_round_of(const HcState &s, const HcInst &in) {
  HcRound r{s.x, s.y, s.theta, s.dt, s.dr, s.failed};
  const int nseg = hc_nseg(in);
  for (int k = 0; k < nseg; ++k) {
    // halving f times is one exact multiplication by 2^-f (f = 0: by 1.0)
    const unsigned f = hc_seg_fail(in, k);
    const double h = hc_pow_half(f);
    r.dt *= h;
    r.dr *= h;
    r.failed += f;
    hc_candidate(r.x, r.y, r.theta, r.dt, r.dr, hc_seg_out(in, k) - 1, &r.x, &r.y, &r.theta);
  }
  const unsigned tf = hc_tail_fail(in);
  const double h = hc_pow_half(tf);
  r.dt *= h;
  r.dr *= h;
  r.failed += tf;
  return r;
}

// a round whose failed counter has reached the limit hands out ONE candidate and ends the chain (Q3:
// has_next() is tested before next() bumps the counter)
HC_HD bool hc_trailing(unsigned failed_at_round, unsigned max_failed) { return failed_at_round >= max_failed; }

// running strict maximum over a round's scores: outcome = 1 + index of the last accepted candidate
// (0 = none), *run = best score afterwards, *nacc = acceptances.  `n` = 6, or 1 for a trailing round.
HC_HD int hc_round_outcome(double enter, const double *s, int n, double *run, int *nacc) {
  double b = enter;
  int out = 0, acc = 0;
  for (int c = 0; c < n; ++c)
    if (b < s[c]) {  // strict: ties are rejections (pose_enumeration_scan_matcher.h:58)
      b = s[c];
      out = c + 1;
      ++acc;
    }
  *run = b;
  *nacc = acc;
  return out;
}

// The same with the decision checked (default mode on the device): `dec` are the scores the comparisons use
// (canonical tree sums, or -- in a re-scored super-step -- the reference's beam-order sums), `hash` identifies a
// pose's vector of beam terms.  A comparison of two tree sums decides like the reference's beam-order sums
// whenever they differ by more than the two orders of summation can (|tree - sequential| <= (n - 1) u sum|t|:
// 2^-40 relative leaves a factor of three for n = 2048) or the term vectors are identical (equal sums in any
// order: a tie, i.e. a rejection, in both).  Anything else is AMBIGUOUS: the super-step is scored again in
// beam order and decided from that.
struct HcDecision {
  int out, nacc, last_acc;  // outcome, acceptances, candidate accepted last (-1: none)
  unsigned accmask;         // bit c: candidate c was accepted
  bool ambiguous;
};
// one comparison of the chain: candidate c with decision score s and term-vector hash h against the running best
// (b, hb), which it replaces when accepted
HC_HD void hc_decide_one(HcDecision &d, double &b, unsigned long long &hb, int c, double s, unsigned long long h,
                         bool check) {
  // (equal fingerprints with DIFFERENT sums cannot be identical term vectors -- those add up to the same bits: a
  // fingerprint collision, as unsettled as differing fingerprints)
  union {
    double d;
    unsigned long long u;
  } sb{s}, bb{b};
  if (check && (h != hb || sb.u != bb.u)) {
    const double diff = s > b ? s - b : b - s;
    const double as = s < 0 ? -s : s, ab = b < 0 ? -b : b;
    if (diff <= (as > ab ? as : ab) * 9.094947017729282e-13) d.ambiguous = true;  // 2^-40 (NaN: a rejection)
  }
  if (b < s) {  // strict: ties are rejections (pose_enumeration_scan_matcher.h:58)
    b = s;
    hb = h;
    d.out = c + 1;
    d.last_acc = c;
    d.accmask |= 1u << c;
    ++d.nacc;
  }
}
HC_HD HcDecision hc_round_decide(double enter, unsigned long long enter_hash, const double *dec,
                                 const unsigned long long *hash, int n, bool check) {
  HcDecision d{0, 0, -1, 0u, false};
  double b = enter;
  unsigned long long hb = enter_hash;
  for (int c = 0; c < n; ++c) hc_decide_one(d, b, hb, c, dec[c], hash[c], check);
  return d;
}

// state at the start of round instance `in` of the tree whose root is (x, y, theta, dt, dr, failed)
HC_HD HcRound hc_round_from(double x, double y, double theta, double dt, double dr, unsigned failed, const HcInst &in) {
  HcState s{};
  s.x = x;
  s.y = y;
  s.theta = theta;
  s.dt = dt;
  s.dr = dr;
  s.failed = failed;
  return hc_round_of(s, in);
}

// What the next super-step's tree hangs off, as a function of the root state, the instance the walk ended in (`in`,
// its round state `r`) and that round's outcome ALONE -- no score enters: the workgroups of a co-resident chain
// tabulate it for every (terminal instance, outcome) pair while the scores are still on their way (hc_resident.hip).
// The acceptance-rate estimate that picks the next tree's shape counts the ACCEPTED ROUNDS on the walked path (the
// path's moves + this round's, if it accepted anything) over its scorer calls, exponentially forgetting
// (MatchJob::consume counts acceptances; r01-r04 did here too, which made the shape a function of the scores: a round
// whose running maximum moved twice counted twice).  The shape only decides what is speculated on, never a result.
struct HcNextCore {
  double x, y, theta, dt, dr;    // best pose = base of the round about to start, and its steps
  double recent_acc, recent_n;
  unsigned failed;
  int shape, done;
  long long batch_calls;         // scorer calls of the walked path
};
HC_HD HcNextCore hc_next_core(const HcState &prev, const HcInst &in, const HcRound &r, int out, unsigned max_failed) {
  HcNextCore n;
  n.x = r.x;
  n.y = r.y;
  n.theta = r.theta;
  n.dt = r.dt;
  n.dr = r.dr;
  n.failed = r.failed;
  if (out > 0) hc_candidate(r.x, r.y, r.theta, r.dt, r.dr, out - 1, &n.x, &n.y, &n.theta);
  const bool trailing = hc_trailing(r.failed, max_failed);
  n.done = trailing ? 1 : 0;
  if (!n.done && out == 0) {
    n.dt = r.dt * 0.5;
    n.dr = r.dr * 0.5;
    n.failed = r.failed + 1;
  }
  n.batch_calls = 6ll * hc_depth(in) + (trailing ? 1 : 6);
  const long long rounds_acc = (long long)hc_nseg(in) + (out > 0 ? 1 : 0);
  n.recent_acc = 0.5 * prev.recent_acc + (double)rounds_acc;
  n.recent_n = 0.5 * prev.recent_n + (double)n.batch_calls;
  n.shape = hc_bucket_of_ratio(n.recent_acc + 0.5, n.recent_n + 4.0);
  return n;
}

// the root state of the next super-step, from the instance the walk ended in (`in`, its round state `r`),
// its outcome and best score.  done: the chain is over.
HC_HD void hc_advance(const HcState &prev, const HcInst &in, const HcRound &r, int out, double run, unsigned max_failed,
                      long long evaluated, HcState *next) {
  const HcNextCore c = hc_next_core(prev, in, r, out, max_failed);
  HcState n = prev;
  n.x = c.x;
  n.y = c.y;
  n.theta = c.theta;
  n.dt = c.dt;
  n.dr = c.dr;
  n.failed = c.failed;
  n.best_prob = run;
  n.done = c.done;
  n.calls = prev.calls + (prev.first ? 1 : 0) + c.batch_calls;
  n.evaluated = prev.evaluated + evaluated;
  n.first = 0;
  n.mode = 0;
  n.steps = prev.steps + 1;
  n.recent_acc = c.recent_acc;
  n.recent_n = c.recent_n;
  n.shape = c.shape;
  *next = n;
}

}  // namespace slamhip
