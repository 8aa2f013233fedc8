// CPU check of the device-resident Monte Carlo (csrc/mc_chain.h): the super-step logic the chain kernel runs --
// candidate j of a state under "all rejected", first acceptance, the enumerator state after the consumed candidates
// (tape position, pending second Marsaglia values, failure counter, halved dispersions) -- executed here over a
// synthetic score function; its trace (candidate poses, scores, accepted flags, final pose) and the number of pairs
// it consumed must equal the plain loop of PoseEnumerationScanMatcher::process_scan
// (src/core/scan_matchers/pose_enumeration_scan_matcher.h:31-77) over GaussianPoseEnumerator (matchers.h, the
// restatement of monte_carlo_scan_matcher.h:10-100 pinned to the reference's MC goldens), bit for bit, over
// consecutive matches on one engine.  Run by tests/test_hc_chain_host.py (also under ASan/UBSan).
#include <cstdio>
#include <cstring>
#include <vector>

#include "matchers.h"
#include "mc_chain.h"

using namespace slamhip;

namespace {

struct Entry {
  double x, y, theta, score;
  long long accepted;
};

struct ScoreFn {
  double tx, ty, tt, quantum;
  double operator()(double x, double y, double th) const {
    const double d2 = (x - tx) * (x - tx) + (y - ty) * (y - ty) + 0.5 * (th - tt) * (th - tt);
    double v = 1.0 / (1.0 + d2) + 0.02 * std::sin(40 * x) * std::cos(31 * y) + 0.01 * std::sin(25 * th);
    if (quantum > 0) v = std::floor(v / quantum) * quantum;
    return v;
  }
};

std::vector<Entry> reference_loop(GaussianPoseEnumerator &pe, const Pose &init, const ScoreFn &f, Pose *best_out) {
  std::vector<Entry> tr;
  Pose best = init;
  double best_prob = f(init.x, init.y, init.theta);
  pe.reset();
  tr.push_back(Entry{init.x, init.y, init.theta, best_prob, 1});
  while (pe.has_next()) {
    const Pose c = pe.next(best);
    const double p = f(c.x, c.y, c.theta);
    const bool ok = best_prob < p;
    pe.feedback(ok);
    tr.push_back(Entry{c.x, c.y, c.theta, p, ok ? 1 : 0});
    if (ok) {
      best = c;
      best_prob = p;
    }
  }
  *best_out = best;
  return tr;
}

// the chain: super-steps of n_slots candidates
std::vector<Entry> chain(GaussianPoseEnumerator &pe, const Pose &init, const ScoreFn &f, int n_slots, Pose *best_out,
                         size_t *pairs_consumed) {
  std::vector<Entry> tr;
  pe.reset();
  const size_t need = 3 * (pe.max_poses() / 2 + 2 + pe.max_poses() / (pe.max_failed() / 3 + 1) + 2) + 8;
  std::vector<double> raw(2 * need);
  pe.copy_tape_abs(pe.tape_pos(), need, raw.data());
  const McPair *tape = reinterpret_cast<const McPair *>(raw.data());
  McState s{};
  s.x = init.x;
  s.y = init.y;
  s.theta = init.theta;
  s.td = pe.base_td();
  s.rd = pe.base_rd();
  s.first = 1;
  while (!s.done) {
    const int avail = (int)mc_available(s, pe.max_failed(), pe.max_poses());
    const int n = avail < n_slots ? avail : n_slots;
    if (s.first) {
      s.best_prob = f(s.x, s.y, s.theta);
      s.calls = 1;
      tr.push_back(Entry{s.x, s.y, s.theta, s.best_prob, 1});
    }
    std::vector<Entry> cand(n);
    int j_acc = -1;
    for (int j = 0; j < n; ++j) {  // "workgroup j"
      mc_candidate(s, tape, j, &cand[j].x, &cand[j].y, &cand[j].theta);
      cand[j].score = f(cand[j].x, cand[j].y, cand[j].theta);
      cand[j].accepted = 0;
    }
    for (int j = 0; j < n; ++j)
      if (s.best_prob < cand[j].score) {
        j_acc = j;
        break;
      }
    const int used = j_acc >= 0 ? j_acc + 1 : n;
    if (j_acc >= 0) cand[j_acc].accepted = 1;
    for (int j = 0; j < used; ++j) tr.push_back(cand[j]);
    const Entry a = j_acc >= 0 ? cand[j_acc] : Entry{0, 0, 0, 0, 0};
    mc_advance(s, tape, n, j_acc, a.x, a.y, a.theta, a.score, 0u, pe.max_failed(), pe.max_poses());
    s.first = 0;
    if ((size_t)s.pos + 3 > need) {
      std::printf("the chain ran past the window: %lld of %zu\n", s.pos, need);
      std::exit(1);
    }
  }
  *best_out = Pose{s.x, s.y, s.theta};
  *pairs_consumed = (size_t)s.pos;
  pe.set_chain_result((size_t)s.pos, s.failed, s.poses, s.td, s.rd, s.has_saved != 0, s.saved);
  return tr;
}

}  // namespace

int main() {
  int matches = 0;
  const unsigned limits[][2] = {{20, 100}, {4096, 4096}, {3, 2}, {30, 1000}, {1, 50}, {7, 7}, {100, 33}};
  for (const auto &lim : limits)
    for (int n_slots : {1, 5, 252, 384})
      for (double quantum : {0.0, 0.01}) {
        GaussianPoseEnumerator ref(1234u + lim[0], 0.2, 0.1, lim[0], lim[1]), dev(1234u + lim[0], 0.2, 0.1, lim[0], lim[1]);
        Pose init{0.3, -0.2, 0.1};
        for (int rep = 0; rep < 3; ++rep) {  // consecutive matches: the engine is never reseeded
          const ScoreFn f{0.5 + 0.1 * rep, 0.1, 0.3, quantum};
          Pose br, bd;
          size_t consumed = 0;
          const size_t pos0 = ref.tape_pos();
          const std::vector<Entry> tr = reference_loop(ref, init, f, &br);
          const std::vector<Entry> td = chain(dev, init, f, n_slots, &bd, &consumed);
          if (tr.size() != td.size() || std::memcmp(tr.data(), td.data(), tr.size() * sizeof(Entry)) != 0 ||
              std::memcmp(&br, &bd, sizeof(Pose)) != 0 || ref.tape_pos() != dev.tape_pos() ||
              ref.tape_pos() - pos0 != consumed) {
            std::printf("MISMATCH limits %u/%u slots %d quantum %g match %d: %zu vs %zu calls, tape %zu vs %zu\n", lim[0],
                        lim[1], n_slots, quantum, rep, tr.size(), td.size(), ref.tape_pos(), dev.tape_pos());
            return 1;
          }
          std::string ka, kb;
          ref.key(ka);
          dev.key(kb);
          if (ka != kb) {
            std::printf("enumerator state differs after the match (limits %u/%u slots %d)\n", lim[0], lim[1], n_slots);
            return 1;
          }
          init = Pose{init.x + 0.01, init.y - 0.02, init.theta + 0.005};
          ++matches;
        }
      }
  std::printf("ok %d matches\n", matches);
  return 0;
}
