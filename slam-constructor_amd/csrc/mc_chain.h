// mc_chain.h -- the Monte-Carlo scan matcher's accept chain on the device: state and closed forms shared by
// the kernel (mc_chain.hip) and the host driver / host test (plain C++).
//
// Reference: MonteCarloScanMatcher + GaussianPoseEnumerator (src/core/scan_matchers/monte_carlo_scan_matcher.h:
// 10-100) driven by PoseEnumerationScanMatcher::process_scan (pose_enumeration_scan_matcher.h:31-77).  A candidate
// is best + (N(0, td), N(0, td), N(0, rd)); the three libstdc++ normal_distributions share one mt19937 and every
// engine word they consume is consumed inside one Marsaglia polar pair, so the k-th pair drawn is a pure function
// of the seed (matchers.h PairTape).  Each distribution hands out the pair's second value on its next call, and
// all three are re-created together (reset_shift), hence they alternate IN PHASE: a "fresh" candidate takes the
// first values of three new pairs (x, y, theta in that order), the next candidate their second values.
//
// One super-step scores the candidates that follow the current state under "every one of them is rejected"
// (they all hang off the same best pose); the replay finds the first one that is accepted -- `best < candidate`,
// strict -- and everything behind it is discarded.  The enumerator state after consuming n candidates is a closed
// form of (tape position, pending second values), so every workgroup derives its own candidate without a loop.
#pragma once

#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#define MC_HD __host__ __device__ __forceinline__
#else
#define MC_HD inline
#endif

namespace slamhip {

struct McPair {
  double ret, saved;  // unit normals in the order a distribution hands them out (PairTape::Pair)
};

struct McState {
  double x, y, theta, best_prob;  // best pose so far and its score
  double td, rd;                  // current dispersions (halved by reset_shift)
  double saved[3];                // pending second values of the x, y, theta distributions (has_saved)
  long long pos;                  // next pair of the tape, relative to the window uploaded for this match
  long long calls, evaluated;
  unsigned failed, poses;
  int has_saved;
  int done, first, mode, steps;
  unsigned long long best_hash;  // fingerprint of the best pose's term vector (checked default mode)
  unsigned rescored;
};

// the three unit normals candidate j (0-based, counted from state s, all earlier ones rejected) is made of
MC_HD void mc_normals(const McState &s, const McPair *tape, int j, double v[3]) {
  if (s.has_saved && j == 0) {
    v[0] = s.saved[0];
    v[1] = s.saved[1];
    v[2] = s.saved[2];
    return;
  }
  const int c = j - (s.has_saved ? 1 : 0);  // candidates after the pending values are used up
  const long long q = s.pos + 3ll * (c >> 1);
  if (c & 1) {
    v[0] = tape[q].saved;
    v[1] = tape[q + 1].saved;
    v[2] = tape[q + 2].saved;
  } else {
    v[0] = tape[q].ret;
    v[1] = tape[q + 1].ret;
    v[2] = tape[q + 2].ret;
  }
}

// GaussianPoseEnumerator::next: best + (ret * stddev + mean) per axis, mean = 0 (the + 0.0 turns -0.0 into +0.0
// like the reference's expression does)
MC_HD void mc_candidate(const McState &s, const McPair *tape, int j, double *x, double *y, double *theta) {
  double v[3];
  mc_normals(s, tape, j, v);
  *x = s.x + (v[0] * s.td + 0.0);
  *y = s.y + (v[1] * s.td + 0.0);
  *theta = s.theta + (v[2] * s.rd + 0.0);
}

// candidates the enumerator still hands out from s if all are rejected: has_next() = failed < max_failed &&
// poses < max_poses (monte_carlo_scan_matcher.h:30-33)
MC_HD unsigned mc_available(const McState &s, unsigned max_failed, unsigned max_poses) {
  if (s.failed >= max_failed || s.poses >= max_poses) return 0u;
  const unsigned a = max_failed - s.failed, b = max_poses - s.poses;
  return a < b ? a : b;
}

// tape position and pending values after n candidates were drawn from s
MC_HD void mc_consume(McState &s, const McPair *tape, int n) {
  if (n <= 0) return;
  int c = n;
  if (s.has_saved) {
    s.has_saved = 0;
    --c;
  }
  s.pos += 3ll * (c >> 1);
  if (c & 1) {
    s.saved[0] = tape[s.pos].saved;
    s.saved[1] = tape[s.pos + 1].saved;
    s.saved[2] = tape[s.pos + 2].saved;
    s.has_saved = 1;
    s.pos += 3;
  }
}

// The state after a super-step of n scored candidates whose first acceptance is candidate j_acc (-1: none):
// feedback(false) j_acc times (or n times), then feedback(true) -- which halves the dispersions and re-creates the
// distributions when more than max_failed / 3 failures preceded it (monte_carlo_scan_matcher.h:44-55,58-70: the
// failure counter is NOT reset by an acceptance below that mark)
MC_HD void mc_advance(McState &s, const McPair *tape, int n, int j_acc, double acc_x, double acc_y, double acc_theta,
                      double acc_prob, unsigned long long acc_hash, unsigned max_failed, unsigned max_poses) {
  const int used = j_acc >= 0 ? j_acc + 1 : n;
  mc_consume(s, tape, used);
  s.calls += used;
  s.poses += (unsigned)used;
  s.failed += (unsigned)(j_acc >= 0 ? j_acc : n);
  if (j_acc >= 0) {
    s.x = acc_x;
    s.y = acc_y;
    s.theta = acc_theta;
    s.best_prob = acc_prob;
    s.best_hash = acc_hash;
    if (s.failed > max_failed / 3) {  // reset_shift(td * 0.5, rd * 0.5)
      s.failed = 0;
      s.td = s.td * 0.5;
      s.rd = s.rd * 0.5;
      s.has_saved = 0;  // fresh distribution objects: a pending second value is dropped
    }
  }
  s.done = (s.failed < max_failed && s.poses < max_poses) ? 0 : 1;
}

}  // namespace slamhip
