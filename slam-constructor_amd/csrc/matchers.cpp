// matchers.cpp -- host drivers that keep the reference's sequential accept/reject semantics while
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

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <memory>
#include <random>
#include <string>
#include <vector>

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
};

template <typename T>
static void put(std::string &out, const T &v) {
  out.append(reinterpret_cast<const char *>(&v), sizeof(T));
}

// The engine's raw output stream does not depend on accept/reject decisions, so speculative
// copies of the Monte-Carlo enumerator share one growing tape of mt19937 words and only carry a
// read position (a 2.5 KB engine copy per tree node would dominate the host time).
struct EngineTape {
  explicit EngineTape(unsigned seed) : engine(seed) {}
  std::mt19937 engine;
  std::vector<std::mt19937::result_type> words;
  size_t base = 0;  // absolute index of words[0]
  std::mt19937::result_type at(size_t i) {
    while (i - base >= words.size()) words.push_back(engine());
    return words[i - base];
  }
  void trim(size_t consumed) {
    if (consumed - base < (1u << 16)) return;
    words.erase(words.begin(), words.begin() + (consumed - base));
    base = consumed;
  }
};

struct TapeEngine {
  std::mt19937::result_type operator()() { return tape->at(pos++); }
  std::shared_ptr<EngineTape> tape;
  size_t pos = 0;
};

// std::generate_canonical<double, 53> over a 32-bit engine: two words, low word first
// (libstdc++ bits/random.tcc)
static inline double canonical(TapeEngine &g) {
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
  double operator()(TapeEngine &g) {
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

// Monte-Carlo: candidate = best + N(0, sigma) per axis from three distributions sharing one
// engine; sigma halves on an acceptance that follows more than max_failed/3 failures (the
// `factor` argument of the reference's reset_shift is ignored there, so it always halves).
class GaussianPoseEnumerator : public PoseEnumerator {
public:
  GaussianPoseEnumerator(unsigned seed, double td, double rd, unsigned max_failed, unsigned max_poses)
      : max_failed_(max_failed), max_poses_(max_poses), base_td_(td), base_rd_(rd) {
    engine_.tape = std::make_shared<EngineTape>(seed);
    reset();
  }
  bool has_next() const override { return failed_ < max_failed_ && poses_ < max_poses_; }
  Pose next(const Pose &prev) override {
    // draw order x, y, theta -- braced-init-list evaluation order in RobotPoseDeltaRV::sample
    const double dx = rv_x_(engine_);
    const double dy = rv_y_(engine_);
    const double dth = rv_t_(engine_);
    return Pose{prev.x + dx, prev.y + dy, prev.theta + dth};
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
    put(out, engine_.pos);
    put(out, failed_);
    put(out, poses_);
    put(out, td_);
    put(out, rd_);
    // a distribution's pending second Marsaglia value is part of the state
    for (const NormalRV *d : {&rv_x_, &rv_y_, &rv_t_}) {
      put(out, d->has_saved);
      if (d->has_saved) put(out, d->saved);
    }
  }
  void trim() override { engine_.tape->trim(engine_.pos); }
  bool cheap_step() const override { return false; }

private:
  void reset_shift(double td, double rd) {
    failed_ = 0;
    td_ = td;
    rd_ = rd;
    // fresh distribution objects: a saved second Marsaglia value is dropped here
    rv_x_ = NormalRV(0, td_);
    rv_y_ = NormalRV(0, td_);
    rv_t_ = NormalRV(0, rd_);
  }
  unsigned max_failed_, max_poses_, failed_ = 0, poses_ = 0;
  double base_td_, base_rd_, td_ = 0, rd_ = 0;
  NormalRV rv_x_, rv_y_, rv_t_;
  TapeEngine engine_;
};

// Hill climbing: rounds of six candidates base +X, -Y, +Th, -X, +Y, -Th (action id % 3 picks the
// axis, id % 2 the sign); a round in which all six were rejected halves both steps and counts as
// failed.  has_next() is checked before next() bumps the failed-round counter, so one trailing
// candidate is evaluated after the last failed round (Q3).  frame rotation is always 0 (Q5).
class HillClimbingPoseEnumerator : public PoseEnumerator {
public:
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
  struct Inst {
    double prio;
    HillClimbingPoseEnumerator st;  // at a round boundary
    Pose best;
    int n_sites;
    int site_node[6], site_branch[6];  // edges to patch with this instance's first node
    bool operator<(const Inst &o) const { return prio < o.prio; }
  };
  std::vector<Inst> heap_;

  void build_rounds(const HillClimbingPoseEnumerator &real, const Pose &best, int budget,
                    double p_accept) {
    heap_.clear();
    Inst r{1.0, real, best, 0, {0}, {0}};
    heap_.push_back(r);
    const double q = 1.0 - p_accept;
    double p_out[7];
    p_out[0] = std::pow(q, 6);
    for (int j = 1; j <= 6; ++j) p_out[j] = p_accept * std::pow(q, 6 - j);
    while (!heap_.empty() && (int)evals.size() + 6 <= budget) {
      // an instance reached with probability P saves ~P round trips (~20 us each) and costs host
      // time plus six evaluations: not worth it below ~1 %
      if (!evals.empty() && heap_.front().prio < 0.01) break;
      std::pop_heap(heap_.begin(), heap_.end());
      Inst in = std::move(heap_.back());
      heap_.pop_back();
      const int first = (int)nodes.size();
      auto patch = [&](int target) {
        if (in.n_sites == 0) root = target;
        for (int s = 0; s < in.n_sites; ++s) nodes[in.site_node[s]].child[in.site_branch[s]] = target;
      };
      if (!in.st.has_next()) {
        patch(kEnd);
        continue;
      }
      HillClimbingPoseEnumerator e_fail = in.st, e_ok = in.st;
      Pose c[6];
      c[0] = e_fail.next(in.best);
      e_fail.feedback(false);
      (void)e_ok.next(in.best);
      e_ok.feedback(true);
      if (!e_fail.has_next()) {
        // trailing candidate after the last failed round (Q3): evaluated, then the loop ends
        evals.push_back(c[0]);
        nodes.push_back(Node{(int)evals.size() - 1, {kEnd, kEnd}});
        patch(first);
        continue;
      }
      for (int k = 1; k < 6; ++k) {
        c[k] = e_fail.next(in.best);
        e_fail.feedback(false);
        (void)e_ok.next(in.best);
        e_ok.feedback(false);
      }
      const int e0 = (int)evals.size();
      for (int k = 0; k < 6; ++k) evals.push_back(c[k]);
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
      const int last = first + 15;  // node(5, j)
      for (int j = 0; j <= 6; ++j) {
        Inst nx{in.prio * p_out[j], j == 0 ? e_fail : e_ok, j == 0 ? in.best : c[j - 1], 0, {0}, {0}};
        if (j < 6) {
          nx.n_sites = 1;
          nx.site_node[0] = last + j;  // candidate 5 rejected, j-1 accepted last
          nx.site_branch[0] = 0;
        } else {
          nx.n_sites = 6;  // candidate 5 accepted, whatever came before
          for (int t = 0; t < 6; ++t) {
            nx.site_node[t] = last + t;
            nx.site_branch[t] = 1;
          }
        }
        heap_.push_back(nx);
        std::push_heap(heap_.begin(), heap_.end());
      }
    }
  }
};

}  // namespace slamhip

struct slamhip_matcher {
  slamhip_ctx *ctx = nullptr;
  slamhip_spe_cfg cfg{};
  std::unique_ptr<slamhip::PoseEnumerator> pe;
  slamhip_observer obs{};
  bool has_obs = false;
  int max_batch = 0;
  double p_accept0 = 0.25;  // prior per-candidate acceptance rate of a fresh match
  slamhip::SpecTree tree;
  long long scorer_calls = 0, poses_evaluated = 0, launches = 0;
  double t_build_us = 0, t_stage_us = 0, t_score_us = 0, t_replay_us = 0;  // last process_scan
};

using namespace slamhip;

namespace {

int invalid_arg(const char *msg) {
  set_error(msg);
  return SLAMHIP_ERR_INVALID;
}

int make_matcher(slamhip_ctx *ctx, const slamhip_spe_cfg *cfg, std::unique_ptr<PoseEnumerator> pe,
                 int default_batch, slamhip_matcher **out) {
  if (!ctx || !cfg || !out) return invalid_arg("null argument");
  auto *m = new slamhip_matcher;
  m->ctx = ctx;
  m->cfg = *cfg;
  m->pe = std::move(pe);
  m->max_batch = default_batch;
  *out = m;
  return SLAMHIP_OK;
}

// carry-in of the GMapping OOPE cache for ONE replayed pose (see gm_carry_fixup in slamhip_api.cpp)
double gm_apply_carry(slamhip_ctx *ctx, int p, double score) {
  GmPoseInfo &gi = ctx->h_gm_info[p];
  double last_v = gi.last_v;
  if (ctx->gm_prob != -1.0 && gi.first_cx == ctx->gm_cx && gi.first_cy == ctx->gm_cy) {
    const double c = ctx->gm_prob;
    if (c != gi.v0) {
      double delta = 0.0;
      for (int b = 0; b < gi.run0_len; ++b)
        delta += (c * ctx->h_weight[b]) * ctx->h_factor[b] - (gi.v0 * ctx->h_weight[b]) * ctx->h_factor[b];
      if (ctx->scan_tot_w != 0.0) score += delta / ctx->scan_tot_w;
    }
    if (gi.last_head == 0) last_v = c;
  }
  ctx->gm_cx = gi.last_cx;
  ctx->gm_cy = gi.last_cy;
  ctx->gm_prob = last_v;
  return score;
}

}  // namespace

extern "C" {

int slamhip_matcher_create_mc(slamhip_ctx *ctx, const slamhip_spe_cfg *cfg, unsigned seed,
                              double td, double rd, unsigned failed_limit, unsigned attempts_limit,
                              slamhip_matcher **out) {
  return make_matcher(ctx, cfg,
                      std::make_unique<GaussianPoseEnumerator>(seed, td, rd, failed_limit, attempts_limit),
                      1024, out);
}

int slamhip_matcher_create_hc(slamhip_ctx *ctx, const slamhip_spe_cfg *cfg, unsigned failed_rounds_limit,
                              double dt, double dr, slamhip_matcher **out) {
  return make_matcher(ctx, cfg, std::make_unique<HillClimbingPoseEnumerator>(failed_rounds_limit, dt, dr),
                      1024, out);
}

int slamhip_matcher_create_bf(slamhip_ctx *ctx, const slamhip_spe_cfg *cfg, const double range9[9],
                              slamhip_matcher **out) {
  if (!range9) return invalid_arg("null range");
  if (!(range9[0] <= range9[1] && range9[3] <= range9[4] && range9[6] <= range9[7]) ||
      !(range9[2] > 0 && range9[5] > 0 && range9[8] > 0))
    return invalid_arg("brute-force ranges need from <= to and positive steps");
  return make_matcher(ctx, cfg, std::make_unique<BruteForcePoseEnumerator>(range9), 8192, out);
}

int slamhip_matcher_destroy(slamhip_matcher *m) {
  delete m;
  return SLAMHIP_OK;
}

int slamhip_matcher_reset_state(slamhip_matcher *m) {
  if (!m) return invalid_arg("null matcher");
  m->pe->reset();
  return SLAMHIP_OK;
}

int slamhip_matcher_set_observer(slamhip_matcher *m, const slamhip_observer *obs) {
  if (!m) return invalid_arg("null matcher");
  m->has_obs = obs != nullptr;
  if (obs) m->obs = *obs;
  return SLAMHIP_OK;
}

int slamhip_matcher_set_batch(slamhip_matcher *m, int max_batch) {
  if (!m || max_batch < 0) return invalid_arg("bad batch");
  if (max_batch > 0) m->max_batch = max_batch;
  return SLAMHIP_OK;
}

int slamhip_matcher_stats(slamhip_matcher *m, long long *scorer_calls, long long *poses_evaluated,
                          long long *launches) {
  if (!m) return invalid_arg("null matcher");
  if (scorer_calls) *scorer_calls = m->scorer_calls;
  if (poses_evaluated) *poses_evaluated = m->poses_evaluated;
  if (launches) *launches = m->launches;
  return SLAMHIP_OK;
}

int slamhip_matcher_timing(slamhip_matcher *m, double *build_us, double *stage_us, double *score_us,
                           double *replay_us) {
  if (!m) return invalid_arg("null matcher");
  if (build_us) *build_us = m->t_build_us;
  if (stage_us) *stage_us = m->t_stage_us;
  if (score_us) *score_us = m->t_score_us;
  if (replay_us) *replay_us = m->t_replay_us;
  return SLAMHIP_OK;
}

int slamhip_matcher_process_scan(slamhip_matcher *m, int map_id, const double init_pose[3],
                                 double out_delta[3], double *out_prob) {
  if (!m || !init_pose || !out_delta || !out_prob) return invalid_arg("null argument");
  slamhip_ctx *ctx = m->ctx;
  SLAMHIP_CHECK(hipSetDevice(ctx->device));
  const bool gm = m->cfg.oope == SLAMHIP_OOPE_GMAPPING;
  m->scorer_calls = m->poses_evaluated = m->launches = 0;
  m->t_build_us = m->t_stage_us = m->t_score_us = m->t_replay_us = 0;
  auto now_us = [] {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
  };
  const int budget = m->max_batch > 0 ? m->max_batch : 256;
  int rc = ensure_pose_capacity(ctx, budget + 1);
  if (rc) return rc;
  m->pe->trim();

  Pose best{init_pose[0], init_pose[1], init_pose[2]};
  double best_prob = 0.0;
  bool first = true;
  // the reference resets the enumerator after scoring the initial pose
  // (pose_enumeration_scan_matcher.h:47); nothing depends on that score, so the initial pose
  // rides in the first speculative batch
  m->pe->reset();
  SpecTree &tree = m->tree;
  double p_accept = m->p_accept0;
  double recent_acc = 0.0, recent_n = 0.0;
  while (true) {
    const int lead = first ? 1 : 0;
    const double t0 = now_us();
    tree.build(*m->pe, best, budget, p_accept);
    const double t1 = now_us();
    m->t_build_us += t1 - t0;
    const int n = lead + (int)tree.evals.size();
    if (n == 0) break;
    double *hp = ctx->h_poses;
    if (first) {
      hp[0] = best.x;
      hp[1] = best.y;
      hp[2] = best.theta;
    }
    for (size_t i = 0; i < tree.evals.size(); ++i) {
      hp[3 * (lead + i)] = tree.evals[i].x;
      hp[3 * (lead + i) + 1] = tree.evals[i].y;
      hp[3 * (lead + i) + 2] = tree.evals[i].theta;
    }
    const double t2 = now_us();
    m->t_stage_us += t2 - t1;
    rc = score_staged(ctx, map_id, &m->cfg, n);
    if (rc) return rc;
    const double t3 = now_us();
    m->t_score_us += t3 - t2;
    m->launches += 1;
    m->poses_evaluated += n;
    const double *sc = ctx->h_scores;
    if (first) {
      best_prob = gm ? gm_apply_carry(ctx, 0, sc[0]) : sc[0];
      m->scorer_calls += 1;
      if (m->has_obs) {
        const double p3[3] = {best.x, best.y, best.theta};
        if (m->obs.on_scan_test) m->obs.on_scan_test(m->obs.user, p3, best_prob);
        if (m->obs.on_pose_update) m->obs.on_pose_update(m->obs.user, p3, best_prob);
      }
      first = false;
    }
    // replay the real enumerator down the tree in the reference's order
    int node = tree.root;
    int batch_n = 0, batch_acc = 0;
    // enumerators that are costly to step (MC) are not re-stepped candidate by candidate: after
    // a fully rejected chain the speculative copy IS the new state; otherwise the real one is
    // fast-forwarded over the walked prefix once
    const bool lazy = !m->pe->cheap_step() && tree.chain_end;
    const Pose best_at_batch_start = best;
    while (node >= 0) {
      const SpecTree::Node &nd = tree.nodes[node];
      Pose c = tree.evals[nd.eval];
      if (!lazy) {
        c = m->pe->next(best);  // same state => the speculated candidate, bit for bit
        if (std::memcmp(&c, &tree.evals[nd.eval], sizeof(Pose)) != 0) {
          set_error("internal: speculated candidate differs from the enumerator's (speculation bug)");
          return SLAMHIP_ERR_STATE;
        }
      }
      double prob = sc[lead + nd.eval];
      if (gm) prob = gm_apply_carry(ctx, lead + nd.eval, prob);
      m->scorer_calls += 1;
      const double p3[3] = {c.x, c.y, c.theta};
      if (m->has_obs && m->obs.on_scan_test) m->obs.on_scan_test(m->obs.user, p3, prob);
      const bool ok = best_prob < prob;  // strict: ties are rejections (Q1)
      if (!lazy) m->pe->feedback(ok);
      ++batch_n;
      if (ok) {
        ++batch_acc;
        best_prob = prob;
        best = c;
        if (m->has_obs && m->obs.on_pose_update) m->obs.on_pose_update(m->obs.user, p3, best_prob);
      }
      node = nd.child[ok ? 1 : 0];
    }
    if (lazy) {
      if (batch_acc == 0 && batch_n == (int)tree.nodes.size()) {
        m->pe = std::move(tree.chain_end);
      } else {
        // a chain walk ends at its first acceptance: batch_n - 1 rejections, then the accepted one
        for (int i = 0; i < batch_n; ++i) {
          (void)m->pe->next(best_at_batch_start);
          m->pe->feedback(batch_acc > 0 && i == batch_n - 1);
        }
      }
    }
    m->t_replay_us += now_us() - t3;
    if (node == SpecTree::kEnd || !m->pe->has_next()) break;
    // acceptance-rate estimate for the next tree (exponentially forgetting)
    recent_acc = 0.5 * recent_acc + batch_acc;
    recent_n = 0.5 * recent_n + batch_n;
    p_accept = std::min(0.5, std::max(0.004, (recent_acc + 0.5) / (recent_n + 4.0)));
  }
  out_delta[0] = best.x - init_pose[0];
  out_delta[1] = best.y - init_pose[1];
  out_delta[2] = best.theta - init_pose[2];
  *out_prob = best_prob;
  if (m->has_obs && m->obs.on_matching_end) m->obs.on_matching_end(m->obs.user, out_delta, best_prob);
  return SLAMHIP_OK;
}

}  // extern "C"
