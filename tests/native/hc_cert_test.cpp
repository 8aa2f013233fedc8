// CPU check of the CERTIFICATE behind the closed-form tail of the co-resident hill-climbing chain (csrc/hc_chain.h
// hc_cert_beam, csrc/hc_resident.hip): an argument about rounding, held here against the plain accept loop of
// PoseEnumerationScanMatcher::process_scan (src/core/scan_matchers/pose_enumeration_scan_matcher.h:31-77) over
// HillClimbingPoseEnumerator, in host arithmetic.  The score is the 1-cell scorer's: per beam, the end point by the
// cached provider's angle addition (score_device.h beam_cell), its cell by a true division and floor (to_cell), a value
// that depends on the CELL alone (a hash of its indices: every cell differs from its neighbours), summed in beam order.
// The "chain" runs the reference's loop round by round and applies the kernel's rule: when a stretch of `tree` rounds
// based on the same pose has failed, the pose's certificate -- the minimum of hc_cert_beam over the beams -- is compared
// with the next round's steps, and if it holds the rest of the match is written in closed form (the candidates' poses,
// the best score, rejected).  The trace must equal the reference loop's, call for call, bit for bit.
// Geometry is adversarial: few beams, levers up to 30 m, end points placed 1e-14 ... 1e-3 m from cell edges, steps
// starting on either side of that distance.  Run by tests/test_hc_chain_host.py (ASan / UBSan).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "hc_chain.h"
#include "matchers.h"

using namespace slamhip;

namespace {

struct Entry {
  double x, y, theta, score;
  long long accepted;
};

struct Beams {
  std::vector<double> r, ca, sa, w;
  double scale, inv_scale;
};

inline int cell_of(double v, double scale) { return (int)std::floor(v / scale); }  // regular_squares_grid.h:40-46

inline double cell_value(int cx, int cy) {  // any function of the cell: a hash, so that neighbours differ
  unsigned long long h = (unsigned long long)(unsigned)cx * 0x9E3779B97F4A7C15ull ^ (unsigned long long)(unsigned)cy * 0xC2B2AE3D27D4EB4Full;
  h ^= h >> 29;
  h *= 0xBF58476D1CE4E5B9ull;
  h ^= h >> 32;
  return (double)(h & 0xfffff) / 1048576.0;
}

double score(const Beams &b, double x, double y, double theta) {
  double sn, cs;
  ::sincos(theta, &sn, &cs);
  double acc = 0.0, tot = 0.0;
  for (size_t i = 0; i < b.r.size(); ++i) {
    const double c = cs * b.ca[i] - sn * b.sa[i];
    const double s = sn * b.ca[i] + cs * b.sa[i];
    const double wx = x + b.r[i] * c, wy = y + b.r[i] * s;
    acc = acc + cell_value(cell_of(wx, b.scale), cell_of(wy, b.scale)) * b.w[i];
    tot += b.w[i];
  }
  return acc / tot;
}

void certificate(const Beams &b, double x, double y, double theta, double *t_t, double *t_r) {
  double sn, cs;
  ::sincos(theta, &sn, &cs);
  double tt = INFINITY, tr = INFINITY;
  for (size_t i = 0; i < b.r.size(); ++i) {
    double t1, t2;
    hc_cert_beam(x, y, sn, cs, std::fabs(theta), b.r[i], b.ca[i], b.sa[i], b.scale, b.inv_scale, &t1, &t2);
    tt = tt < t1 ? tt : t1;
    tr = tr < t2 ? tr : t2;
  }
  *t_t = tt;
  *t_r = tr;
}

std::vector<Entry> reference_loop(unsigned max_failed, double dt, double dr, const Pose &init, const Beams &b) {
  std::vector<Entry> tr;
  HillClimbingPoseEnumerator pe(max_failed, dt, dr);
  Pose best = init;
  double best_prob = score(b, init.x, init.y, init.theta);
  tr.push_back(Entry{init.x, init.y, init.theta, best_prob, 1});
  pe.reset();
  while (pe.has_next()) {
    const Pose c = pe.next(best);
    const double p = score(b, c.x, c.y, c.theta);
    const bool ok = best_prob < p;
    pe.feedback(ok);
    tr.push_back(Entry{c.x, c.y, c.theta, p, ok ? 1 : 0});
    if (ok) {
      best_prob = p;
      best = c;
    }
  }
  return tr;
}

// the same loop, round by round, with the kernel's rule: after `tree` consecutive failed rounds on one base pose (a
// super-step whose walk accepted nothing) the pose's certificate is held against the NEXT round's steps
std::vector<Entry> certified_loop(unsigned max_failed, double dt0, double dr0, const Pose &init, const Beams &b, int tree,
                                  long long *tail_out) {
  std::vector<Entry> tr;
  Pose best = init;
  double best_prob = score(b, init.x, init.y, init.theta);
  tr.push_back(Entry{init.x, init.y, init.theta, best_prob, 1});
  unsigned failed = 0;
  double dt = dt0, dr = dr0;
  int fails_on_base = 0;
  *tail_out = 0;
  for (;;) {
    // (has_next() is tested before next() bumps the counter: a round at the limit hands out one candidate)
    const bool trailing = failed >= max_failed;
    const Pose base = best;
    bool round_failed = true;
    for (int c = 0; c < (trailing ? 1 : 6); ++c) {
      Entry e;
      hc_candidate(base.x, base.y, base.theta, dt, dr, c, &e.x, &e.y, &e.theta);
      e.score = score(b, e.x, e.y, e.theta);
      e.accepted = best_prob < e.score ? 1 : 0;
      if (e.accepted) {
        best_prob = e.score;
        best = Pose{e.x, e.y, e.theta};
        round_failed = false;
      }
      tr.push_back(e);
    }
    if (trailing) break;
    if (round_failed) {
      dt *= 0.5;
      dr *= 0.5;
      ++failed;
      ++fails_on_base;
    } else {
      fails_on_base = 0;
    }
    if (round_failed && fails_on_base % tree == 0 && failed < max_failed) {
      double t_t, t_r;
      certificate(b, best.x, best.y, best.theta, &t_t, &t_r);
      const bool inert = hc_inert(best.x, best.y, best.theta, dt, dr);
      if (inert || (dt < t_t && dr < t_r)) {
        const long long tail = 6ll * (long long)(max_failed - failed) + 1ll;
        for (long long q = 0; q < tail; ++q) {
          const double hlf = hc_pow_half((unsigned)(q / 6));
          Entry e;
          hc_candidate(best.x, best.y, best.theta, dt * hlf, dr * hlf, (int)(q % 6), &e.x, &e.y, &e.theta);
          e.score = best_prob;
          e.accepted = 0;
          tr.push_back(e);
        }
        *tail_out = tail;
        break;
      }
    }
  }
  return tr;
}

}  // namespace

int main(int argc, char **argv) {
  const long long n_matches = argc > 1 ? std::atoll(argv[1]) : 20000;
  std::mt19937_64 rng(2024);
  std::uniform_real_distribution<double> u(0.0, 1.0);
  long long tails = 0, certified_early = 0, calls = 0, late_accepts = 0;
  for (long long it = 0; it < n_matches; ++it) {
    Beams b;
    b.scale = (it % 3 == 0) ? 0.05 : ((it % 3 == 1) ? 0.1 : 0.025);
    b.inv_scale = 1.0 / b.scale;
    const int nb = 1 + (int)(u(rng) * 12);
    const Pose init{(u(rng) - 0.5) * (it % 5 == 0 ? 200.0 : 8.0), (u(rng) - 0.5) * (it % 5 == 0 ? 200.0 : 8.0), (u(rng) - 0.5) * 6.2};
    for (int i = 0; i < nb; ++i) {
      const double a = (u(rng) - 0.5) * 4.6;
      double r = 0.3 + u(rng) * (i == 0 && it % 2 == 0 ? 30.0 : 10.0);
      const double c = std::cos(init.theta + a), s = std::sin(init.theta + a);
      if (i < 2) {  // this beam's end point a distance d from a cell edge, on either side, in x or in y
        const double d = std::pow(10.0, -14.0 + 11.0 * u(rng)) * (u(rng) < 0.5 ? -1.0 : 1.0);
        if (u(rng) < 0.5 && std::fabs(c) > 0.2) {
          const double edge = std::round((init.x + r * c) / b.scale) * b.scale;
          r = std::fabs((edge + d - init.x) / c);
        } else if (std::fabs(s) > 0.2) {
          const double edge = std::round((init.y + r * s) / b.scale) * b.scale;
          r = std::fabs((edge + d - init.y) / s);
        }
      }
      b.r.push_back(r);
      b.ca.push_back(std::cos(a));
      b.sa.push_back(std::sin(a));
      b.w.push_back(1.0 / nb);
    }
    const unsigned max_failed = (it % 4 == 0) ? 60 : 128;
    const double dt = std::pow(10.0, -9.0 + 8.0 * u(rng)), dr = std::pow(10.0, -9.0 + 8.0 * u(rng));
    const int tree = 1 + (int)(u(rng) * 42);
    const auto ref = reference_loop(max_failed, dt, dr, init, b);
    long long tail = 0;
    const auto got = certified_loop(max_failed, dt, dr, init, b, tree, &tail);
    if (ref.size() != got.size() || std::memcmp(ref.data(), got.data(), ref.size() * sizeof(Entry)) != 0) {
      std::printf("FAIL: match %lld (%d beams, scale %g, limit %u, steps %g %g, tree %d): %zu reference calls, %zu certified\n", it, nb,
                  b.scale, max_failed, dt, dr, tree, ref.size(), got.size());
      for (size_t i = 0; i < std::min(ref.size(), got.size()); ++i)
        if (std::memcmp(&ref[i], &got[i], sizeof(Entry)) != 0) {
          std::printf("  first difference at call %zu: ref (%.17g %.17g %.17g) %.17g %lld, certified (%.17g %.17g %.17g) %.17g %lld\n", i,
                      ref[i].x, ref[i].y, ref[i].theta, ref[i].score, ref[i].accepted, got[i].x, got[i].y, got[i].theta,
                      got[i].score, got[i].accepted);
          break;
        }
      return 1;
    }
    calls += (long long)ref.size();
    tails += tail > 0;
    certified_early += tail > 6 * 75 + 1 && max_failed == 128;  // (the identical-pose rule alone starts at failed round ~50)
    for (size_t i = ref.size() > 400 ? ref.size() - 400 : 0; i < ref.size(); ++i) late_accepts += ref[i].accepted;
  }
  std::printf("ok %lld matches, %lld scorer calls; closed-form tails in %lld, before the identical-pose rule could in %lld; "
              "%lld acceptances among the last 400 calls of a match\n", n_matches, calls, tails, certified_early, late_accepts);
  return 0;
}
