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
// Speculation: between two acceptances neither the best pose nor the
// enumerator's step/dispersion changes, so the next K candidates can be generated from a COPY of
// the enumerator under the assumption "all rejected", scored in one launch, and then replayed on
// the real enumerator in order with the running best score.  The replay stops trusting the batch
// at the first acceptance (HC: at the end of that round, because a round's six candidates all
// derive from the base pose latched at the round's first next()).  Observers therefore see
// exactly the reference's on_scan_test / on_pose_update sequence; discarded evaluations are never
// reported and never touch the GMapping OOPE cache.

#include <cmath>
#include <cstring>
#include <memory>
#include <random>
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
  // true while candidates speculated BEFORE an acceptance are still the ones the enumerator
  // would hand out AFTER it
  virtual bool speculation_survives_accept() const { return false; }
};

// Monte-Carlo: candidate = best + N(0, sigma) per axis from three distributions sharing one
// engine; sigma halves on an acceptance that follows more than max_failed/3 failures (the
// `factor` argument of the reference's reset_shift is ignored there, so it always halves).
class GaussianPoseEnumerator : public PoseEnumerator {
public:
  GaussianPoseEnumerator(unsigned seed, double td, double rd, unsigned max_failed, unsigned max_poses)
      : max_failed_(max_failed), max_poses_(max_poses), base_td_(td), base_rd_(rd), engine_(seed) {
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

private:
  void reset_shift(double td, double rd) {
    failed_ = 0;
    td_ = td;
    rd_ = rd;
    // fresh distribution objects: a saved second Marsaglia value is dropped here
    rv_x_ = std::normal_distribution<>(0, td_);
    rv_y_ = std::normal_distribution<>(0, td_);
    rv_t_ = std::normal_distribution<>(0, rd_);
  }
  unsigned max_failed_, max_poses_, failed_ = 0, poses_ = 0;
  double base_td_, base_rd_, td_ = 0, rd_ = 0;
  std::normal_distribution<> rv_x_, rv_y_, rv_t_;
  std::mt19937 engine_;
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
  bool speculation_survives_accept() const override { return action_id_ < 6; }

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
  bool speculation_survives_accept() const override { return true; }

private:
  double r_[9];
  double x_ = 0, y_ = 0, t_ = 0;
  bool base_set_ = false;
  Pose base_{0, 0, 0};
};

}  // namespace slamhip

struct slamhip_matcher {
  slamhip_ctx *ctx = nullptr;
  slamhip_spe_cfg cfg{};
  std::unique_ptr<slamhip::PoseEnumerator> pe;
  slamhip_observer obs{};
  bool has_obs = false;
  int max_batch = 0;
  long long scorer_calls = 0, poses_evaluated = 0, launches = 0;
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
                      256, out);
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

int slamhip_matcher_process_scan(slamhip_matcher *m, int map_id, const double init_pose[3],
                                 double out_delta[3], double *out_prob) {
  if (!m || !init_pose || !out_delta || !out_prob) return invalid_arg("null argument");
  slamhip_ctx *ctx = m->ctx;
  SLAMHIP_CHECK(hipSetDevice(ctx->device));
  const bool gm = m->cfg.oope == SLAMHIP_OOPE_GMAPPING;
  m->scorer_calls = m->poses_evaluated = m->launches = 0;
  const int K = m->max_batch > 0 ? m->max_batch : 256;
  int rc = ensure_pose_capacity(ctx, K + 1);
  if (rc) return rc;

  Pose best{init_pose[0], init_pose[1], init_pose[2]};
  double best_prob = 0.0;
  bool first = true;
  // the reference resets the enumerator after scoring the initial pose
  // (pose_enumeration_scan_matcher.h:47); nothing depends on that score, so the initial pose
  // rides in the first speculative batch
  m->pe->reset();
  std::vector<Pose> cands;
  cands.reserve(K);
  while (true) {
    cands.clear();
    {
      auto spec = m->pe->clone();
      while ((int)cands.size() < K && spec->has_next()) {
        cands.push_back(spec->next(best));
        spec->feedback(false);
      }
    }
    const int lead = first ? 1 : 0;
    const int n = lead + (int)cands.size();
    if (n == 0) break;
    double *hp = ctx->h_poses;
    if (first) {
      hp[0] = best.x;
      hp[1] = best.y;
      hp[2] = best.theta;
    }
    for (size_t i = 0; i < cands.size(); ++i) {
      hp[3 * (lead + i)] = cands[i].x;
      hp[3 * (lead + i) + 1] = cands[i].y;
      hp[3 * (lead + i) + 2] = cands[i].theta;
    }
    rc = score_staged(ctx, map_id, &m->cfg, n);
    if (rc) return rc;
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
    if (cands.empty()) break;
    bool accepted_any = false;
    for (size_t i = 0; i < cands.size(); ++i) {
      if (accepted_any && !m->pe->speculation_survives_accept()) break;
      // the real enumerator hands out the speculated candidate (same state, same draws)
      const Pose c = m->pe->next(best);
      double prob = sc[lead + i];
      if (gm) prob = gm_apply_carry(ctx, lead + (int)i, prob);
      m->scorer_calls += 1;
      const double p3[3] = {c.x, c.y, c.theta};
      if (m->has_obs && m->obs.on_scan_test) m->obs.on_scan_test(m->obs.user, p3, prob);
      const bool ok = best_prob < prob;  // strict: ties are rejections (Q1)
      m->pe->feedback(ok);
      if (!ok) continue;
      best_prob = prob;
      best = c;
      accepted_any = true;
      if (m->has_obs && m->obs.on_pose_update) m->obs.on_pose_update(m->obs.user, p3, best_prob);
    }
  }
  out_delta[0] = best.x - init_pose[0];
  out_delta[1] = best.y - init_pose[1];
  out_delta[2] = best.theta - init_pose[2];
  *out_prob = best_prob;
  if (m->has_obs && m->obs.on_matching_end) m->obs.on_matching_end(m->obs.user, out_delta, best_prob);
  return SLAMHIP_OK;
}

}  // extern "C"
