// matchers.h -- host drivers that keep the reference's sequential accept/reject semantics while
// the scoring runs in speculative GPU batches.
//
// Reference behaviour restated here (paths relative to the reference root):
//   PoseEnumerationScanMatcher::process_scan   src/core/scan_matchers/pose_enumeration_scan_matcher.h:31-77
//   GaussianPoseEnumerator (MC)                src/core/scan_matchers/monte_carlo_scan_matcher.h:10-82
//   Distorsion1DPoseEnumerator +
//   FailedRoundsLimitedPoseEnumerator (HC)     src/core/scan_matchers/hill_climbing_scan_matcher.h:10-126
//   BruteForcePoseEnumerator (BF)              src/core/scan_matchers/brute_force_scan_matcher.h:10-64
//
// Speculation.  The accept/reject chain is a walk down a binary decision tree:
// at every node the enumerator (a small state machine) hands out one candidate, and the two
// children are the enumerator states after feedback(false) / feedback(true) with the
// corresponding best pose.  Every node's candidate is known WITHOUT any score, so a launch
// evaluates a whole sub-tree at once: nodes are expanded best-first by path probability (an
// adaptive per-candidate acceptance rate), identical enumerator states are merged (the tree is a
// DAG: inside an HC round only "which candidate was accepted last" matters), and bitwise
// identical poses share one GPU evaluation.  The host then replays the real enumerator down the
// tree in the reference's order with the running best score until it walks off the expanded part.
// A round trip costs ~20 us while 1024 extra poses cost ~5 us, so trading launches for speculative
// poses is the right exchange on this machine.  Observers see exactly the reference's
// on_scan_test / on_pose_update sequence; evaluations off the taken path are never reported and
// never touch the GMapping OOPE cache.

#pragma once

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <memory>
#include <random>
#include <string>
#include <vector>

#include "mt_block.h"

#include "slamhip_internal.h"

namespace slamhip {

struct Pose {
  double x, y, theta;
};

class PoseEnumerator {
public:
  virtual ~PoseEnumerator() = default;
  virtual bool has_next() const = 0;
  virtual Pose next(const Pose &prev) = 0;
  virtual void reset() = 0;
  virtual void feedback(bool ok) = 0;
  virtual std::unique_ptr<PoseEnumerator> clone() const = 0;
  // raw bytes identifying the state (two states with equal keys enumerate identically)
  virtual void key(std::string &out) const = 0;
  // false when feedback(true) and feedback(false) lead to the same future (brute force)
  virtual bool accept_changes_future() const { return true; }
  // housekeeping between matches (drop consumed random words)
  virtual void trim() {}
  // false when next() is costly (MC draws three normals): the replay then avoids re-stepping
  virtual bool cheap_step() const { return true; }
  // become a copy of `other` (same dynamic type)
  virtual void assign(const PoseEnumerator &other) = 0;
  // called while a batch is on the GPU: work that does not depend on its outcome
  virtual void idle_work() {}
};

template <typename T>
static void put(std::string &out, const T &v) {
  out.append(reinterpret_cast<const char *>(&v), sizeof(T));
}

// std::generate_canonical<double, 53> over a 32-bit engine: two words, low word first
// (libstdc++ bits/random.tcc)
template <typename Engine>
static inline double canonical(Engine &g) {
  double sum = 0.0, tmp = 1.0;
  for (int k = 2; k != 0; --k) {
    sum += double(g()) * tmp;
    tmp *= 4294967296.0;
  }
  double ret = sum / tmp;
  if (ret >= 1.0) ret = std::nextafter(1.0, 0.0);
  return ret;
}

// std::normal_distribution<double>: Marsaglia polar with one saved value (libstdc++
// bits/random.tcc).  Restated with its state in the open so speculative copies can be compared
// and keyed; the golden MC traces pin the stream against the reference's libstdc++.
struct NormalRV {
  double mean = 0, stddev = 1, saved = 0;
  bool has_saved = false;
  NormalRV() = default;
  NormalRV(double m, double s) : mean(m), stddev(s) {}
  template <typename Engine>
  double operator()(Engine &g) {
    double ret;
    if (has_saved) {
      has_saved = false;
      ret = saved;
    } else {
      double x, y, r2;
      do {
        x = 2.0 * canonical(g) - 1.0;
        y = 2.0 * canonical(g) - 1.0;
        r2 = x * x + y * y;
      } while (r2 > 1.0 || r2 == 0.0);
      const double mult = std::sqrt(-2 * std::log(r2) / r2);
      saved = x * mult;
      has_saved = true;
      ret = y * mult;
    }
    return ret * stddev + mean;
  }
};

// The Monte-Carlo enumerator's three distributions share one engine, and every engine word they
// consume is consumed inside one Marsaglia polar pair.  The k-th pair drawn -- by whichever
// distribution, whatever the accept/reject history, whether or not a reset_shift dropped a saved value
// in between -- is therefore a pure function of the seed.  The pairs (a log and a sqrt each) live on a
// shared tape: computed once, read by every speculative copy of the enumerator through an index (a
// 2.5 KB engine copy per tree node would dominate the host time), and filled AHEAD by the matcher
// while it waits for the GPU.
struct PairTape {
  explicit PairTape(unsigned seed) : engine(seed) {}  // (the output sequence of std::mt19937(seed): mt_block.cpp)
  struct Pair {
    double ret, saved;  // unit normals in the order the distribution hands them out: y*mult, then x*mult
  };
  Mt19937Block engine;
  std::vector<Pair> pairs;
  size_t base = 0;  // absolute index of pairs[0]
  // One block of the engine = 624 words = 156 attempts of the polar method (four words each, accepted or not), so
  // the tape grows a block at a time: the attempts' arithmetic is vectorized (mt_block.cpp), the accepted ones --
  // 0 < r2 <= 1, pi / 4 of them -- take the logarithm in order.  Generating ahead of the consumer changes nothing:
  // the tape is a function of the seed alone.
  void generate() {
    constexpr int kAttempts = 624 / 4;
    double x[kAttempts], y[kAttempts], r2[kAttempts];
    polar_attempts(engine.next_block(), kAttempts, x, y, r2);
    for (int a = 0; a < kAttempts; ++a) {
      if (r2[a] > 1.0 || r2[a] == 0.0) continue;
      const double mult = std::sqrt(-2 * std::log(r2[a]) / r2[a]);
      pairs.push_back(Pair{y[a] * mult, x[a] * mult});
    }
  }
  const Pair &at(size_t i) {
    while (i - base >= pairs.size()) generate();
    return pairs[i - base];
  }
  // make sure pairs up to absolute index `upto` exist, about `max_new` new ones per call (whole blocks)
  void prefetch(size_t upto, int max_new) {
    while (max_new > 0 && base + pairs.size() < upto) {
      const size_t before = pairs.size();
      generate();
      max_new -= (int)(pairs.size() - before);
    }
  }
  void trim(size_t consumed) {
    if (consumed - base < (1u << 15)) return;
    pairs.erase(pairs.begin(), pairs.begin() + (consumed - base));
    base = consumed;
  }
};

// NormalRV over the pair tape: same hand-out order and the same `ret * stddev + mean`
struct TapeNormal {
  double mean = 0, stddev = 1, saved = 0;
  bool has_saved = false;
  TapeNormal() = default;
  TapeNormal(double m, double s) : mean(m), stddev(s) {}
  double draw(PairTape &tape, size_t &pos) {
    double ret;
    if (has_saved) {
      has_saved = false;
      ret = saved;
    } else {
      const PairTape::Pair &p = tape.at(pos++);
      saved = p.saved;
      has_saved = true;
      ret = p.ret;
    }
    return ret * stddev + mean;
  }
};

// Monte-Carlo: candidate = best + N(0, sigma) per axis from three distributions sharing one
// engine; sigma halves on an acceptance that follows more than max_failed/3 failures (the
// `factor` argument of the reference's reset_shift is ignored there, so it always halves).
class GaussianPoseEnumerator : public PoseEnumerator {
public:
  GaussianPoseEnumerator(unsigned seed, double td, double rd, unsigned max_failed, unsigned max_poses)
      : max_failed_(max_failed), max_poses_(max_poses), base_td_(td), base_rd_(rd) {
    tape_ = std::make_shared<PairTape>(seed);
    reset();
  }
  bool has_next() const override { return failed_ < max_failed_ && poses_ < max_poses_; }
  Pose next(const Pose &prev) override {
    // draw order x, y, theta -- braced-init-list evaluation order in RobotPoseDeltaRV::sample
    const double dx = rv_x_.draw(*tape_, pos_);
    const double dy = rv_y_.draw(*tape_, pos_);
    const double dth = rv_t_.draw(*tape_, pos_);
    return Pose{prev.x + dx, prev.y + dy, prev.theta + dth};
  }
  // host idle time (the GPU is scoring): polar pairs for the candidates still to come -- two poses
  // take three pairs
  void idle_work() override {
    const size_t left = max_poses_ > poses_ ? max_poses_ - poses_ : 0;
    tape_->prefetch(pos_ + (3 * left) / 2 + 8, 384);
  }
  void reset() override {
    poses_ = 0;
    reset_shift(base_td_, base_rd_);
  }
  void feedback(bool ok) override {
    ++poses_;
    if (!ok) {
      ++failed_;
      return;
    }
    if (failed_ <= max_failed_ / 3) return;
    reset_shift(td_ * 0.5, rd_ * 0.5);
  }
  std::unique_ptr<PoseEnumerator> clone() const override {
    return std::make_unique<GaussianPoseEnumerator>(*this);
  }
  void key(std::string &out) const override {
    put(out, pos_);
    put(out, failed_);
    put(out, poses_);
    put(out, td_);
    put(out, rd_);
    // a distribution's pending second Marsaglia value is part of the state
    for (const TapeNormal *d : {&rv_x_, &rv_y_, &rv_t_}) {
      put(out, d->has_saved);
      if (d->has_saved) put(out, d->saved);
    }
  }
  void trim() override { tape_->trim(pos_); }
  bool cheap_step() const override { return false; }
  void assign(const PoseEnumerator &o) override { *this = static_cast<const GaussianPoseEnumerator &>(o); }

  // ---- the device chain (mc_chain.hip) runs the enumerator itself: what it starts from and what it hands back
  unsigned max_failed() const { return max_failed_; }
  unsigned max_poses() const { return max_poses_; }
  double base_td() const { return base_td_; }
  double base_rd() const { return base_rd_; }
  size_t tape_pos() const { return pos_; }
  // host idle time while a chain runs on the GPU: pairs for the NEXT match
  void prefetch_ahead(size_t pairs_from_pos, int max_new) { tape_->prefetch(pos_ + pairs_from_pos, max_new); }
  // pairs [from, from + n) of the tape by absolute position (generated on demand; from >= the trimmed base)
  void copy_tape_abs(size_t from, size_t n, double *ret_saved_pairs) {
    for (size_t i = 0; i < n; ++i) {
      const PairTape::Pair &p = tape_->at(from + i);
      ret_saved_pairs[2 * i] = p.ret;
      ret_saved_pairs[2 * i + 1] = p.saved;
    }
  }
  size_t tape_generated_upto() const { return tape_->base + tape_->pairs.size(); }
  void set_chain_result(size_t pairs_consumed, unsigned failed, unsigned poses, double td, double rd, bool has_saved,
                        const double saved[3]) {
    pos_ += pairs_consumed;
    failed_ = failed;
    poses_ = poses;
    td_ = td;
    rd_ = rd;
    rv_x_ = TapeNormal(0, td_);
    rv_y_ = TapeNormal(0, td_);
    rv_t_ = TapeNormal(0, rd_);
    if (has_saved) {
      rv_x_.has_saved = rv_y_.has_saved = rv_t_.has_saved = true;
      rv_x_.saved = saved[0];
      rv_y_.saved = saved[1];
      rv_t_.saved = saved[2];
    }
  }

private:
  void reset_shift(double td, double rd) {
    failed_ = 0;
    td_ = td;
    rd_ = rd;
    // fresh distribution objects: a saved second Marsaglia value is dropped here
    rv_x_ = TapeNormal(0, td_);
    rv_y_ = TapeNormal(0, td_);
    rv_t_ = TapeNormal(0, rd_);
  }
  unsigned max_failed_, max_poses_, failed_ = 0, poses_ = 0;
  double base_td_, base_rd_, td_ = 0, rd_ = 0;
  TapeNormal rv_x_, rv_y_, rv_t_;
  std::shared_ptr<PairTape> tape_;
  size_t pos_ = 0;  // next pair of the tape
};

// Hill climbing: rounds of six candidates base +X, -Y, +Th, -X, +Y, -Th (action id % 3 picks the
// axis, id % 2 the sign); a round in which all six were rejected halves both steps and counts as
// failed.  has_next() is checked before next() bumps the failed-round counter, so one trailing
// candidate is evaluated after the last failed round (Q3).  frame rotation is always 0 (Q5).
class HillClimbingPoseEnumerator : public PoseEnumerator {
public:
  HillClimbingPoseEnumerator() : HillClimbingPoseEnumerator(0, 0, 0) {}
  HillClimbingPoseEnumerator(unsigned max_failed_rounds, double dt, double dr)
      : max_failed_rounds_(max_failed_rounds), base_dt_(dt), base_dr_(dr) {
    reset();
  }
  bool has_next() const override { return failed_rounds_ < max_failed_rounds_; }
  Pose next(const Pose &prev) override {
    if (action_id_ >= 6) {
      if (round_failed_) {
        dt_ *= 0.5;
        dr_ *= 0.5;
        ++failed_rounds_;
      }
      reset_round();
    }
    if (!base_set_) {
      base_ = prev;
      base_set_ = true;
    }
    Pose p = base_;
    const double dir = (action_id_ % 2) ? -1 : 1;
    const double fcos = std::cos(0.0), fsin = std::sin(0.0);
    switch (action_id_ % 3) {
      case 0:
        p.x += fcos * dir * dt_;
        p.y += fsin * dir * dt_;
        break;
      case 1:
        p.x += -fsin * dir * dt_;
        p.y += fcos * dir * dt_;
        break;
      default:
        p.theta += dir * dr_;
        break;
    }
    ++action_id_;
    return p;
  }
  void reset() override {
    failed_rounds_ = 0;
    dt_ = base_dt_;
    dr_ = base_dr_;
    reset_round();
  }
  void feedback(bool ok) override { round_failed_ = round_failed_ && !ok; }
  std::unique_ptr<PoseEnumerator> clone() const override {
    return std::make_unique<HillClimbingPoseEnumerator>(*this);
  }
  void key(std::string &out) const override {
    put(out, failed_rounds_);
    put(out, dt_);
    put(out, dr_);
    put(out, action_id_);
    put(out, base_set_);
    put(out, round_failed_);
    if (base_set_) put(out, base_);
  }
  void assign(const PoseEnumerator &o) override { *this = static_cast<const HillClimbingPoseEnumerator &>(o); }
  // true before the first candidate of a round (fresh, or all six of the previous round handed out)
  bool at_round_boundary() const { return action_id_ >= 6 || (action_id_ == 0 && !base_set_); }

private:
  void reset_round() {
    action_id_ = 0;
    base_set_ = false;
    round_failed_ = true;
  }
  unsigned max_failed_rounds_, failed_rounds_ = 0;
  double base_dt_, base_dr_, dt_ = 0, dr_ = 0;
  unsigned action_id_ = 0;
  bool base_set_ = false, round_failed_ = true;
  Pose base_{0, 0, 0};
};

// Brute force: x fastest, then y, then theta; offsets accumulate by += step; the base pose is
// latched at the first next() and never cleared (not even by reset()).
class BruteForcePoseEnumerator : public PoseEnumerator {
public:
  explicit BruteForcePoseEnumerator(const double r[9]) {
    std::memcpy(r_, r, sizeof(r_));
    reset();
  }
  bool has_next() const override { return t_ <= r_[7]; }
  Pose next(const Pose &prev) override {
    if (!base_set_) {
      base_ = prev;
      base_set_ = true;
    }
    return Pose{base_.x + x_, base_.y + y_, base_.theta + t_};
  }
  void reset() override {
    x_ = r_[0];
    y_ = r_[3];
    t_ = r_[6];
  }
  void feedback(bool) override {
    if (x_ < r_[1]) {
      x_ += r_[2];
      return;
    }
    x_ = r_[0];
    if (y_ < r_[4]) {
      y_ += r_[5];
      return;
    }
    y_ = r_[3];
    t_ += r_[8];
  }
  std::unique_ptr<PoseEnumerator> clone() const override {
    return std::make_unique<BruteForcePoseEnumerator>(*this);
  }
  void key(std::string &out) const override {
    put(out, x_);
    put(out, y_);
    put(out, t_);
    put(out, base_set_);
    if (base_set_) put(out, base_);
  }
  bool accept_changes_future() const override { return false; }
  void assign(const PoseEnumerator &o) override { *this = static_cast<const BruteForcePoseEnumerator &>(o); }
  // the base pose, latched by the first next() of the enumerator's life and never cleared (the reference's reset()
  // leaves _base_pose_is_set alone, brute_force_scan_matcher.h:27-40): later matches enumerate around it
  const Pose &latch_base(const Pose &first_prev) {
    if (!base_set_) {
      base_ = first_prev;
      base_set_ = true;
    }
    return base_;
  }

private:
  double r_[9];
  double x_ = 0, y_ = 0, t_ = 0;
  bool base_set_ = false;
  Pose base_{0, 0, 0};
};

}  // namespace slamhip

namespace slamhip {

// Speculation DAG (see the file header).  Nodes are candidate evaluations; child[0] / child[1] is
// where the walk continues after a rejection / an acceptance.
struct SpecTree {
  static constexpr int kUnexpanded = -1, kEnd = -2;
  struct Node {
    int eval;      // index of this node's candidate in `evals`
    int child[2];
  };
  std::vector<Node> nodes;
  std::vector<Pose> evals;
  int root = kUnexpanded;
  // chain builds only: the enumerator state after every chain candidate was rejected
  std::unique_ptr<PoseEnumerator> chain_end;

  // reach probability below which a round instance is not worth speculating: a lone matcher trades
  // ~20 us round trips against cheap evaluations (1 %); a filter that shares every launch among
  // all its particles pays per evaluation instead and wants a much higher bar
  double min_reach = 0.02;

  void build(const PoseEnumerator &real, const Pose &best, int budget, double p_accept) {
    nodes.clear();
    evals.clear();
    root = kUnexpanded;
    chain_end.reset();
    const auto *hc = dynamic_cast<const HillClimbingPoseEnumerator *>(&real);
    if (hc && budget >= 6 && hc->at_round_boundary())
      build_rounds(*hc, best, budget, p_accept);
    else
      build_chain(real, best, budget, p_accept);
  }

private:
  // Generic: one chain under the assumption "every candidate is rejected".  Its expected useful
  // length is ~1/p_accept, so the chain is cut at a few times that (generating candidates nobody
  // replays costs host time: MC draws three normals per candidate).
  void build_chain(const PoseEnumerator &real, const Pose &best, int budget, double p_accept) {
    auto st = real.clone();
    const bool same_future = !st->accept_changes_future();
    int depth = budget;
    // with a geometric run length the cost per useful candidate (round trip + wasted host work)
    // is minimal near 2 / p_accept
    if (!same_future) depth = std::min(budget, std::max(16, (int)(2.0 / p_accept)));
    int prev = -1;
    chain_end.reset();
    struct KeepEnd {
      SpecTree *t;
      std::unique_ptr<PoseEnumerator> &st;
      ~KeepEnd() { t->chain_end = std::move(st); }
    } keep{this, st};
    while ((int)nodes.size() < depth) {
      if (!st->has_next()) {
        link(prev, kEnd, same_future);
        return;
      }
      const Pose c = st->next(best);
      st->feedback(false);
      const int node = (int)nodes.size();
      evals.push_back(c);
      nodes.push_back(Node{node, {kUnexpanded, kUnexpanded}});
      link(prev, node, same_future);
      prev = node;
    }
  }
  void link(int prev, int node, bool same_future) {
    if (prev < 0) {
      root = node;
      return;
    }
    nodes[prev].child[0] = node;
    if (same_future) nodes[prev].child[1] = node;
  }

  // Hill climbing: whole rounds.  A round from a given boundary state has six fixed candidates and
  // seven outcomes (none accepted, or candidate j accepted last); in-round node (k, j) = "about to
  // evaluate candidate k, candidate j-1 accepted last (j = 0: none)".  Round instances are expanded
  // best-first by outcome probability until the evaluation budget is spent.
  // expanded round instance: the two possible end-of-round enumerator states, its six candidates
  // and where its nodes start; children are (parent, outcome) pairs, so the heap moves 16 bytes
  struct Round {
    HillClimbingPoseEnumerator e_fail, e_ok;
    Pose best, c[6];
    int first;
    int outcome;  // the outcome of the PARENT round that led here (-1: root)
  };
  struct Cand {
    double prio;
    int parent, outcome;
    bool operator<(const Cand &o) const { return prio < o.prio; }
  };
  std::vector<Round> rounds_;
  std::vector<Cand> heap_;

  void build_rounds(const HillClimbingPoseEnumerator &real, const Pose &best, int budget,
                    double p_accept) {
    heap_.clear();
    rounds_.clear();
    heap_.push_back(Cand{1.0, -1, -1});
    const double q = 1.0 - p_accept;
    // outcome probabilities of a round (only speculation priorities: products instead of seven pow()
    // calls per plan, which were a third of the filter's planning time)
    double qk[7], p_out[7];
    qk[0] = 1.0;
    for (int k = 1; k <= 6; ++k) qk[k] = qk[k - 1] * q;
    p_out[0] = qk[6];
    for (int j = 1; j <= 6; ++j) p_out[j] = p_accept * qk[6 - j];
    while (!heap_.empty() && (int)evals.size() + 6 <= budget) {
      // an instance reached with probability P saves ~P round trips (~20 us each) and costs host
      // time plus six evaluations: not worth it below min_reach
      if (!evals.empty() && heap_.front().prio < min_reach) break;
      std::pop_heap(heap_.begin(), heap_.end());
      const Cand cd = heap_.back();
      heap_.pop_back();
      const HillClimbingPoseEnumerator *st = &real;
      Pose in_best = best;
      if (cd.parent >= 0) {
        const Round &pr = rounds_[cd.parent];
        st = cd.outcome == 0 ? &pr.e_fail : &pr.e_ok;
        in_best = cd.outcome == 0 ? pr.best : pr.c[cd.outcome - 1];
      }
      const int first = (int)nodes.size();
      auto patch = [&](int target) {
        if (cd.parent < 0) {
          root = target;
          return;
        }
        const int last = rounds_[cd.parent].first + 15;  // node(5, j) of the parent round
        if (cd.outcome < 6) {
          nodes[last + cd.outcome].child[0] = target;  // candidate 5 rejected, outcome-1 accepted last
        } else {
          for (int t = 0; t < 6; ++t) nodes[last + t].child[1] = target;  // candidate 5 accepted
        }
      };
      if (!st->has_next()) {
        patch(kEnd);
        continue;
      }
      // `st` may point INTO rounds_: copy the state before emplace_back can reallocate the vector
      // (a dangling read here made the plan -- and, through a garbage enumerator, the accept chain --
      // depend on what the allocator had done with the freed block)
      const HillClimbingPoseEnumerator from = *st;
      rounds_.emplace_back();
      Round &rd = rounds_.back();
      rd.e_fail = from;
      rd.e_ok = from;
      rd.best = in_best;
      rd.first = first;
      rd.outcome = cd.outcome;
      rd.c[0] = rd.e_fail.next(in_best);
      rd.e_fail.feedback(false);
      (void)rd.e_ok.next(in_best);
      rd.e_ok.feedback(true);
      if (!rd.e_fail.has_next()) {
        // trailing candidate after the last failed round (Q3): evaluated, then the loop ends
        evals.push_back(rd.c[0]);
        nodes.push_back(Node{(int)evals.size() - 1, {kEnd, kEnd}});
        patch(first);
        rounds_.pop_back();
        continue;
      }
      for (int k = 1; k < 6; ++k) {
        rd.c[k] = rd.e_fail.next(in_best);
        rd.e_fail.feedback(false);
        (void)rd.e_ok.next(in_best);
        rd.e_ok.feedback(false);
      }
      const int e0 = (int)evals.size();
      for (int k = 0; k < 6; ++k) evals.push_back(rd.c[k]);
      // node(k, j) -> first + k(k+1)/2 + j,  j in [0, k]
      for (int k = 0; k < 6; ++k)
        for (int j = 0; j <= k; ++j) {
          Node nd{e0 + k, {kUnexpanded, kUnexpanded}};
          if (k < 5) {
            const int nb = first + (k + 1) * (k + 2) / 2;
            nd.child[0] = nb + j;
            nd.child[1] = nb + k + 1;
          }
          nodes.push_back(nd);
        }
      patch(first);
      const int me = (int)rounds_.size() - 1;
      // a hill climb keeps moving the way it just moved: the outcome that repeats the parent's
      // move gets `repeat_boost` times its share (renormalised)
      double w[7], tot = 0;
      for (int j = 0; j <= 6; ++j) {
        w[j] = p_out[j] * ((j > 0 && j == cd.outcome) ? repeat_boost : 1.0);
        tot += w[j];
      }
      for (int j = 0; j <= 6; ++j) {
        const double prio = cd.prio * w[j] / tot;
        if (prio < min_reach) continue;  // would never be popped (the loop stops below min_reach)
        heap_.push_back(Cand{prio, me, j});
        std::push_heap(heap_.begin(), heap_.end());
      }
    }
  }

public:
  double repeat_boost = 1.0;
};

}  // namespace slamhip

namespace slamhip {

// GMapping OOPE cache carried from call to call (gmapping_occupancy_observation_pe.h:43-44)
struct GmCarry {
  int cx = 0, cy = 0;
  double prob = -1.0;
};

// carry-in of the cache for ONE replayed pose: a pose whose first beam lands in the cell the
// previous call ended in re-uses the cached value for its whole first run (Q19)
static double gm_apply_carry(GmCarry &cr, const GmPoseInfo &gi, double score, const slamhip_ctx *ctx) {
  double last_v = gi.last_v;
  if (cr.prob != -1.0 && gi.first_cx == cr.cx && gi.first_cy == cr.cy) {
    const double c = cr.prob;
    if (c != gi.v0) {
      double delta = 0.0;
      for (int b = 0; b < gi.run0_len; ++b)
        delta += (c * ctx->h_weight[b]) * ctx->h_factor[b] - (gi.v0 * ctx->h_weight[b]) * ctx->h_factor[b];
      if (ctx->scan_tot_w != 0.0) score += delta / ctx->scan_tot_w;
    }
    if (gi.last_head == 0) last_v = c;
  }
  cr.cx = gi.last_cx;
  cr.cy = gi.last_cy;
  cr.prob = last_v;
  return score;
}

// One process_scan as a resumable state machine: plan() emits the poses of the next speculative
// batch, consume() replays the accept chain over their scores.  A matcher drives one job per
// launch; the GMapping filter drives all its particles' jobs in lock-step through shared launches.
class MatchJob {
public:
  PoseEnumerator *pe = nullptr;  // not owned
  bool gm = false;
  const slamhip_observer *obs = nullptr;
  Pose init{0, 0, 0}, best{0, 0, 0};
  double best_prob = 0.0;
  bool first = true, done = false;
  GmCarry carry;
  GmCarry carry_in;  // what the job started from
  GmPoseInfo first_info{};  // side outputs of the initial pose (cross-particle carry check)
  double first_raw_score = 0.0;
  long long scorer_calls = 0, poses_evaluated = 0, launches = 0;
  double t_build_us = 0, t_replay_us = 0;
  bool timed = true;  // the filter's hundred jobs per round switch the clock reads off (4 per job and round)
  SpecTree tree;

  void start(PoseEnumerator *e, const Pose &init_pose, bool gmapping, const slamhip_observer *o,
             const GmCarry &carry_in, double p_accept0) {
    pe = e;
    gm = gmapping;
    obs = o;
    init = best = init_pose;
    best_prob = 0.0;
    first = true;
    done = false;
    carry = carry_in;
    this->carry_in = carry_in;
    scorer_calls = poses_evaluated = launches = 0;
    t_build_us = t_replay_us = 0;
    p_accept_ = p_accept0;
    recent_acc_ = recent_n_ = 0.0;
    pe->trim();
    // the reference resets the enumerator after scoring the initial pose
    // (pose_enumeration_scan_matcher.h:47); nothing depends on that score, so the initial pose
    // rides in the first speculative batch
    pe->reset();
  }

  // writes the batch (x, y, theta triples) and returns its size; 0 = nothing left to evaluate
  int plan(int budget, double *out) {
    if (done) return 0;
    const double t0 = timed ? now_us() : 0.0;
    lead_ = first ? 1 : 0;
    tree.build(*pe, best, budget, p_accept_);
    const int n = lead_ + (int)tree.evals.size();
    if (n == 0) {
      done = true;
      return 0;
    }
    if (first) {
      out[0] = best.x;
      out[1] = best.y;
      out[2] = best.theta;
    }
    for (size_t i = 0; i < tree.evals.size(); ++i) {
      out[3 * (lead_ + i)] = tree.evals[i].x;
      out[3 * (lead_ + i) + 1] = tree.evals[i].y;
      out[3 * (lead_ + i) + 2] = tree.evals[i].theta;
    }
    if (timed) t_build_us += now_us() - t0;
    planned_ = n;
    return n;
  }

  // Checked default mode (fp = the batch's term-vector fingerprints): would the walk over this batch meet a
  // `best < candidate` that the canonical tree sums cannot settle -- the two scores within 2^-40 (relative) of
  // each other, more than the two orders of summation can differ by, and the term vectors not identical?
  // Nothing is consumed; the caller then scores the batch once more in beam order and hands those sums to
  // consume() as `dec`.  (csrc/hc_chain.h hc_decide_one: the same test on the device chain.)
  bool ambiguous(const double *sc, const unsigned long long *fp) const {
    double b = first ? sc[0] : best_prob;
    unsigned long long hb = first ? fp[0] : best_fp;
    int node = tree.root;
    while (node >= 0) {
      const SpecTree::Node &nd = tree.nodes[node];
      const double s = sc[lead_ + nd.eval];
      const unsigned long long h = fp[lead_ + nd.eval];
      // (equal fingerprints with DIFFERENT sums cannot be identical term vectors -- those add up to the same bits --
      // so that is a fingerprint collision and as unsettled as differing fingerprints)
      if ((h != hb || std::memcmp(&s, &b, sizeof(double)) != 0) &&
          std::fabs(s - b) <= std::max(std::fabs(s), std::fabs(b)) * 9.094947017729282e-13)
        return true;
      const bool ok = b < s;
      if (ok) {
        b = s;
        hb = h;
      }
      node = nd.child[ok ? 1 : 0];
    }
    return false;
  }
  // Checked default mode over the GMapping OOPE (r06): would the walk over this batch meet a `best < candidate` whose
  // two scores -- the cache applied in call order, on a copy -- lie within 2^-40 (relative) of each other?  The canonical
  // sums with the device's exp cannot settle such a comparison the way the reference's beam-order sums with glibc's exp
  // would (two zero scores are sums of zeros: settled).  Nothing is consumed.
  bool gm_unsettled(const double *sc, const GmPoseInfo *gi, const slamhip_ctx *ctx) const {
    GmCarry c = carry;
    double b = first ? gm_apply_carry(c, gi[0], sc[0], ctx) : best_prob;
    int node = tree.root;
    while (node >= 0) {
      const SpecTree::Node &nd = tree.nodes[node];
      const double s = gm_apply_carry(c, gi[lead_ + nd.eval], sc[lead_ + nd.eval], ctx);
      if (std::fabs(s - b) <= std::max(std::fabs(s), std::fabs(b)) * 9.094947017729282e-13 && !(s == 0.0 && b == 0.0)) return true;
      const bool ok = b < s;
      if (ok) b = s;
      node = nd.child[ok ? 1 : 0];
    }
    return false;
  }
  int planned() const { return planned_; }
  unsigned long long best_fp = 0;  // fingerprint of the best pose's term vector (checked default mode)

  // sc / gi point at this job's slice of the batch results; fp (optional): the fingerprints, remembered for the
  // best pose; dec (optional): the scores the comparisons are decided from -- the beam-order sums of a batch
  // scored twice, dec_best the current best pose's -- while sc stays what is stored and reported
  int consume(const double *sc, const GmPoseInfo *gi, const slamhip_ctx *ctx, const unsigned long long *fp = nullptr,
              const double *dec = nullptr, double dec_best = 0.0) {
    const double t0 = timed ? now_us() : 0.0;
    launches += 1;
    poses_evaluated += planned_;
    if (first) {
      if (fp) best_fp = fp[0];
      if (dec) dec_best = dec[0];
      if (gm) {
        first_info = gi[0];
        first_raw_score = sc[0];
      }
      best_prob = gm ? gm_apply_carry(carry, gi[0], sc[0], ctx) : sc[0];
      scorer_calls += 1;
      if (obs) {
        const double p3[3] = {best.x, best.y, best.theta};
        if (obs->on_scan_test) obs->on_scan_test(obs->user, p3, best_prob);
        if (obs->on_pose_update) obs->on_pose_update(obs->user, p3, best_prob);
      }
      first = false;
    }
    // replay the real enumerator down the tree in the reference's order
    int node = tree.root;
    int batch_n = 0, batch_acc = 0;
    // enumerators that are costly to step (MC) are not re-stepped candidate by candidate: after
    // a fully rejected chain the speculative copy IS the new state; otherwise the real one is
    // fast-forwarded over the walked prefix once
    const bool lazy = !pe->cheap_step() && tree.chain_end;
    const Pose best_at_batch_start = best;
    while (node >= 0) {
      const SpecTree::Node &nd = tree.nodes[node];
      Pose c = tree.evals[nd.eval];
      if (!lazy) {
        c = pe->next(best);  // same state => the speculated candidate, bit for bit
        if (std::memcmp(&c, &tree.evals[nd.eval], sizeof(Pose)) != 0) {
          set_error("internal: speculated candidate differs from the enumerator's (speculation bug)");
          return SLAMHIP_ERR_STATE;
        }
      }
      double prob = sc[lead_ + nd.eval];
      if (gm) prob = gm_apply_carry(carry, gi[lead_ + nd.eval], prob, ctx);
      scorer_calls += 1;
      const double p3[3] = {c.x, c.y, c.theta};
      if (obs && obs->on_scan_test) obs->on_scan_test(obs->user, p3, prob);
      const bool ok = dec ? dec_best < dec[lead_ + nd.eval] : best_prob < prob;  // strict: ties are rejections (Q1)
      if (!lazy) pe->feedback(ok);
      ++batch_n;
      if (ok) {
        ++batch_acc;
        if (dec) dec_best = dec[lead_ + nd.eval];
        if (fp) best_fp = fp[lead_ + nd.eval];
        best_prob = prob;
        best = c;
        if (obs && obs->on_pose_update) obs->on_pose_update(obs->user, p3, best_prob);
      }
      node = nd.child[ok ? 1 : 0];
    }
    if (lazy) {
      if (batch_acc == 0 && batch_n == (int)tree.nodes.size()) {
        pe->assign(*tree.chain_end);
      } else {
        // a chain walk ends at its first acceptance: batch_n - 1 rejections, then the accepted one
        for (int i = 0; i < batch_n; ++i) {
          (void)pe->next(best_at_batch_start);
          pe->feedback(batch_acc > 0 && i == batch_n - 1);
        }
      }
    }
    if (timed) t_replay_us += now_us() - t0;
    if (node == SpecTree::kEnd || !pe->has_next()) {
      done = true;
      return SLAMHIP_OK;
    }
    // acceptance-rate estimate for the next tree (exponentially forgetting)
    recent_acc_ = 0.5 * recent_acc_ + batch_acc;
    recent_n_ = 0.5 * recent_n_ + batch_n;
    p_accept_ = std::min(0.5, std::max(0.004, (recent_acc_ + 0.5) / (recent_n_ + 4.0)));
    return SLAMHIP_OK;
  }

  void delta(double out[3]) const {
    out[0] = best.x - init.x;
    out[1] = best.y - init.y;
    out[2] = best.theta - init.theta;
  }

  double p_accept() const { return p_accept_; }

  static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
  }

private:
  double p_accept_ = 0.25, recent_acc_ = 0, recent_n_ = 0;
  int lead_ = 0, planned_ = 0;
};

}  // namespace slamhip

