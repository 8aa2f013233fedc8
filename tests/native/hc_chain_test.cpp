// CPU check of the device-resident hill climbing (csrc/hc_chain.h, hc_shape.h): the super-step logic the
// chain kernel runs (state of a round instance from the root state, outcome of a round, which instances
// lie on the path, the next root state) is executed here lane by lane -- ballots become loops -- over a
// synthetic score function, and its trace (candidate poses, scores, accepted flags, final pose) must equal
// the plain loop of PoseEnumerationScanMatcher::process_scan
// (src/core/scan_matchers/pose_enumeration_scan_matcher.h:31-77) over HillClimbingPoseEnumerator
// (matchers.h), bit for bit.  Run by tests/test_hc_chain_host.py (also under ASan/UBSan).
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "hc_shape.h"
#include "matchers.h"

using namespace slamhip;

namespace {

struct Entry {
  double x, y, theta, score;
  long long accepted;  // no padding: traces are compared with memcmp
};

// deterministic "map": a bumpy bowl around a target pose, quantised so that ties happen
struct ScoreFn {
  double tx, ty, tt, quantum;
  double operator()(double x, double y, double th) const {
    const double d2 = (x - tx) * (x - tx) + (y - ty) * (y - ty) + 0.5 * (th - tt) * (th - tt);
    double v = 1.0 / (1.0 + d2) + 0.02 * std::sin(40 * x) * std::cos(31 * y) + 0.01 * std::sin(25 * th);
    if (quantum > 0) v = std::floor(v / quantum) * quantum;
    return v;
  }
};

std::vector<Entry> reference_loop(unsigned max_failed, double dt, double dr, const Pose &init, const ScoreFn &f,
                                  Pose *best_out, double *prob_out) {
  std::vector<Entry> tr;
  HillClimbingPoseEnumerator pe(max_failed, dt, dr);
  Pose best = init;
  double best_prob = f(init.x, init.y, init.theta);
  tr.push_back(Entry{init.x, init.y, init.theta, best_prob, 1});
  pe.reset();
  while (pe.has_next()) {
    const Pose c = pe.next(best);
    const double p = f(c.x, c.y, c.theta);
    const bool ok = best_prob < p;
    pe.feedback(ok);
    tr.push_back(Entry{c.x, c.y, c.theta, p, ok ? 1 : 0});
    if (ok) {
      best_prob = p;
      best = c;
    }
  }
  *best_out = best;
  *prob_out = best_prob;
  return tr;
}

// one process_scan the way the chain of kernels runs it
std::vector<Entry> chain_loop(const std::vector<HcShape> &shapes, unsigned max_failed, double dt, double dr,
                              const Pose &init, const ScoreFn &f, double p_accept0, Pose *best_out, double *prob_out,
                              int *steps_out, long long *evaluated_out, bool inert_tail = false,
                              long long *tail_calls_out = nullptr) {
  std::vector<Entry> tr;
  HcState st{};
  st.x = init.x;
  st.y = init.y;
  st.theta = init.theta;
  st.dt = dt;
  st.dr = dr;
  st.shape = hc_bucket_of(p_accept0);
  st.first = 1;
  std::vector<double> sc(kHcSlots, -777.0);  // a slot the kernel skips keeps garbage
  for (int k = 0; k < 100000 && !st.done; ++k) {
    const HcShape &sh = shapes[st.shape];
    // body of kernel k: every slot scores its pose
    for (int slot = 0; slot < kHcSlots; ++slot) {
      if (slot == kHcSlots - 1) {
        if (st.first) sc[slot] = f(st.x, st.y, st.theta);
        continue;
      }
      const int i = slot / 6, c = slot % 6;
      if (i >= sh.n_inst) continue;
      const HcInst &in = sh.inst[i];
      if (!hc_is_root(in) && st.failed + hc_nfail_parent(in) >= max_failed) continue;
      const HcRound r = hc_round_of(st, in);
      if (hc_trailing(r.failed, max_failed) && c > 0) continue;
      double px, py, pth;
      hc_candidate(r.x, r.y, r.theta, r.dt, r.dr, c, &px, &py, &pth);
      sc[slot] = f(px, py, pth);
    }
    // prologue of kernel k+1: the replay
    const double root_prob = st.first ? sc[kHcSlots - 1] : st.best_prob;
    int out[kHcMaxInst], nacc[kHcMaxInst];
    double run[kHcMaxInst], enter[kHcMaxInst];
    bool reach[kHcMaxInst], trailing[kHcMaxInst], valid[kHcMaxInst];
    for (int i = 0; i < kHcMaxInst; ++i) {
      const HcInst &me = sh.inst[i];
      const bool active = i < sh.n_inst;
      reach[i] = active && (hc_is_root(me) || st.failed + hc_nfail_parent(me) < max_failed);
      trailing[i] = reach[i] && hc_trailing(st.failed + hc_nfail(me), max_failed);
      enter[i] = (!active || hc_bp_inst(me) < 0) ? root_prob : sc[6 * hc_bp_inst(me) + hc_bp_cand(me)];
      out[i] = hc_round_outcome(enter[i], &sc[6 * i], trailing[i] ? 1 : 6, &run[i], &nacc[i]);
    }
    unsigned long long has[7] = {0};
    for (int i = 0; i < kHcMaxInst; ++i)
      if (reach[i]) has[out[i]] |= 1ull << i;
    int terminal = -1, n_term = 0;
    for (int i = 0; i < kHcMaxInst; ++i) {
      const HcInst &me = sh.inst[i];
      valid[i] = reach[i];
      for (int o = 0; o < 7; ++o) valid[i] = valid[i] && (me.w[o] & ~has[o]) == 0ull;
      if (valid[i] && (trailing[i] || hc_child(me, out[i]) < 0)) {
        terminal = i;
        ++n_term;
      }
    }
    if (n_term != 1) {
      std::printf("FAIL: %d terminal rounds in super-step %d\n", n_term, k);
      std::exit(1);
    }
    // trace of the walked rounds, by depth
    if (st.first) tr.push_back(Entry{st.x, st.y, st.theta, root_prob, 1});
    const size_t base = tr.size();
    const HcInst &tm = sh.inst[terminal];
    tr.resize(base + 6 * hc_depth(tm) + (trailing[terminal] ? 1 : 6));
    for (int i = 0; i < kHcMaxInst; ++i) {
      if (!valid[i]) continue;
      const HcInst &me = sh.inst[i];
      const HcRound r = hc_round_of(st, me);
      double b = enter[i];
      for (int c = 0; c < (trailing[i] ? 1 : 6); ++c) {
        Entry e;
        hc_candidate(r.x, r.y, r.theta, r.dt, r.dr, c, &e.x, &e.y, &e.theta);
        e.score = sc[6 * i + c];
        e.accepted = b < e.score ? 1 : 0;
        if (e.accepted) b = e.score;
        tr[base + 6 * hc_depth(me) + c] = e;
      }
    }
    HcState next;
    const HcRound r = hc_round_of(st, tm);
    hc_advance(st, tm, r, out[terminal], run[terminal], max_failed, 6ll * sh.n_inst + (st.first ? 1 : 0), &next);
    st = next;
    // r06: the inert tail (hc_inert; hc_resident.hip tabulates it per next root): every candidate of every further round
    // IS the root pose -- the chain ends, the remaining 6 x (limit - failed) + 1 scorer calls are written in closed form
    if (inert_tail && !st.done && hc_inert(st.x, st.y, st.theta, st.dt, st.dr)) {
      const long long tail = 6ll * (long long)(max_failed - st.failed) + 1ll;
      for (long long q = 0; q < tail; ++q) {
        const double hlf = hc_pow_half((unsigned)(q / 6));
        Entry e;
        hc_candidate(st.x, st.y, st.theta, st.dt * hlf, st.dr * hlf, (int)(q % 6), &e.x, &e.y, &e.theta);
        e.score = st.best_prob;
        e.accepted = 0;
        tr.push_back(e);
      }
      st.calls += tail;
      st.done = 1;
      if (tail_calls_out) *tail_calls_out = tail;
    }
  }
  *best_out = Pose{st.x, st.y, st.theta};
  *prob_out = st.best_prob;
  *steps_out = st.steps;
  *evaluated_out = st.evaluated;
  if ((size_t)st.calls != tr.size()) {
    std::printf("FAIL: state counts %lld calls, trace holds %zu\n", st.calls, tr.size());
    std::exit(1);
  }
  return tr;
}

}  // namespace

int main() {
  std::mt19937 rng(11);
  std::uniform_real_distribution<double> u(-1.0, 1.0);
  long long cases = 0, calls = 0, steps = 0, evaluated = 0, tails = 0;
  for (int variant = 0; variant < 4; ++variant) {
    // shapes as the matcher builds them, plus small / boosted ones (a small shape walks off early and often)
    std::vector<HcShape> shapes(kHcShapes);
    const double boost = variant == 1 ? 3.0 : 1.0;
    const int max_inst = variant == 2 ? 5 : (variant == 3 ? 17 : kHcMaxInst);
    for (int b = 0; b < kHcShapes; ++b) hc_build_shape(hc_bucket_rate(b), boost, 0.002, max_inst, &shapes[b]);
    for (int rep = 0; rep < 60; ++rep) {
      const unsigned max_failed = rep % 5 == 0 ? 1 : (rep % 5 == 1 ? 6 : (rep % 5 == 2 ? 20 : (rep % 5 == 3 ? 128 : 250)));
      const double dt = rep % 3 == 0 ? 0.1 : 0.37, dr = rep % 2 ? 0.1 : 0.013;
      const Pose init{3.0 * u(rng), 3.0 * u(rng), u(rng)};
      const ScoreFn f{init.x + 0.4 * u(rng), init.y + 0.4 * u(rng), init.theta + 0.2 * u(rng),
                      rep % 4 == 0 ? 1e-3 : (rep % 4 == 1 ? 1e-6 : 0.0)};
      Pose b0, b1;
      double p0, p1;
      int st;
      long long ev;
      const auto ref = reference_loop(max_failed, dt, dr, init, f, &b0, &p0);
      const auto got = chain_loop(shapes, max_failed, dt, dr, init, f, 0.004 + 0.25 * (u(rng) + 1) * 0.5, &b1, &p1, &st, &ev);
      if (ref.size() != got.size() || std::memcmp(ref.data(), got.data(), ref.size() * sizeof(Entry)) != 0 ||
          std::memcmp(&b0, &b1, sizeof(Pose)) != 0 || std::memcmp(&p0, &p1, sizeof(double)) != 0) {
        std::printf("FAIL: variant %d case %d (max_failed %u): %zu reference calls, %zu chain calls\n", variant, rep,
                    max_failed, ref.size(), got.size());
        for (size_t i = 0; i < std::min(ref.size(), got.size()); ++i)
          if (std::memcmp(&ref[i], &got[i], sizeof(Entry)) != 0) {
            std::printf("  first difference at call %zu: ref (%.17g %.17g %.17g) %.17g %d, chain (%.17g %.17g %.17g) %.17g %d\n",
                        i, ref[i].x, ref[i].y, ref[i].theta, ref[i].score, ref[i].accepted, got[i].x, got[i].y,
                        got[i].theta, got[i].score, got[i].accepted);
            break;
          }
        return 1;
      }
      // ... and with the inert tail in closed form: the same trace, fewer super-steps where the limit lies behind the
      // point at which the steps vanish (0.1 x 2^-k against poses of a few metres: k ~ 52 ... 57)
      {
        Pose b2;
        double p2;
        int st2;
        long long ev2, tail = 0;
        const auto got2 = chain_loop(shapes, max_failed, dt, dr, init, f, 0.1, &b2, &p2, &st2, &ev2, true, &tail);
        if (ref.size() != got2.size() || std::memcmp(ref.data(), got2.data(), ref.size() * sizeof(Entry)) != 0 ||
            std::memcmp(&b0, &b2, sizeof(Pose)) != 0 || std::memcmp(&p0, &p2, sizeof(double)) != 0) {
          std::printf("FAIL: inert tail, variant %d case %d (max_failed %u): %zu reference calls, %zu chain calls\n", variant,
                      rep, max_failed, ref.size(), got2.size());
          return 1;
        }
        if (max_failed >= 128 && (tail < 6 + 1 || ev2 >= ev)) {
          std::printf("FAIL: inert tail not taken, variant %d case %d (max_failed %u): tail %lld\n", variant, rep, max_failed, tail);
          return 1;
        }
        tails += tail > 0;
      }
      ++cases;
      calls += (long long)ref.size();
      steps += st;
      evaluated += ev;
    }
  }
  std::printf("ok %lld matches, %lld scorer calls in %lld super-steps (%.1f calls per step, %.2f evaluations per call)\n",
              cases, calls, steps, (double)calls / steps, (double)evaluated / calls);
  std::printf("inert tails taken in %lld of them, same traces\n", tails);
  return 0;
}
