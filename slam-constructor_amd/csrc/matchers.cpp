// matchers.cpp -- C-ABI of the matcher tier (include/slamhip.h): MC / HC / BF process_scan over the
// speculative drivers of matchers.h.

#include <cstdlib>

#include "matchers.h"

struct slamhip_matcher {
  slamhip_ctx *ctx = nullptr;
  slamhip_spe_cfg cfg{};
  std::unique_ptr<slamhip::PoseEnumerator> pe;
  slamhip_observer obs{};
  bool has_obs = false;
  int max_batch = 0;
  double p_accept0 = 0.25;  // prior per-candidate acceptance rate of a fresh match
  slamhip::MatchJob job;
  double t_stage_us = 0, t_score_us = 0;
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
  if (scorer_calls) *scorer_calls = m->job.scorer_calls;
  if (poses_evaluated) *poses_evaluated = m->job.poses_evaluated;
  if (launches) *launches = m->job.launches;
  return SLAMHIP_OK;
}

int slamhip_matcher_timing(slamhip_matcher *m, double *build_us, double *stage_us, double *score_us,
                           double *replay_us) {
  if (!m) return invalid_arg("null matcher");
  if (build_us) *build_us = m->job.t_build_us;
  if (stage_us) *stage_us = m->t_stage_us;
  if (score_us) *score_us = m->t_score_us;
  if (replay_us) *replay_us = m->job.t_replay_us;
  return SLAMHIP_OK;
}

int slamhip_matcher_process_scan(slamhip_matcher *m, int map_id, const double init_pose[3],
                                 double out_delta[3], double *out_prob) {
  if (!m || !init_pose || !out_delta || !out_prob) return invalid_arg("null argument");
  slamhip_ctx *ctx = m->ctx;
  SLAMHIP_CHECK(hipSetDevice(ctx->device));
  const bool gm = m->cfg.oope == SLAMHIP_OOPE_GMAPPING;
  const int budget = m->max_batch > 0 ? m->max_batch : 256;
  int rc = ensure_pose_capacity(ctx, budget + 1);
  if (rc) return rc;
  m->t_stage_us = m->t_score_us = 0;
  GmCarry carry;
  carry.cx = ctx->gm_cx;
  carry.cy = ctx->gm_cy;
  carry.prob = ctx->gm_prob;
  MatchJob &job = m->job;
  if (const char *e = getenv("SLAMHIP_HC_BOOST")) job.tree.repeat_boost = atof(e);
  if (const char *e = getenv("SLAMHIP_MIN_REACH")) job.tree.min_reach = atof(e);
  job.start(m->pe.get(), Pose{init_pose[0], init_pose[1], init_pose[2]}, gm, m->has_obs ? &m->obs : nullptr,
            carry, m->p_accept0);
  while (!job.done) {
    const int n = job.plan(budget, ctx->h_poses);
    if (n == 0) break;
    const double t0 = MatchJob::now_us();
    unsigned seq = 0;
    rc = score_staged(ctx, map_id, &m->cfg, n, nullptr, 0, &seq);
    if (rc) return rc;
    m->pe->idle_work();  // outcome-independent host work while the batch is on the GPU (MC: polar pairs)
    rc = score_wait(ctx, seq);
    if (rc) return rc;
    m->t_score_us += MatchJob::now_us() - t0;
    rc = job.consume(ctx->h_scores, ctx->h_gm_info, ctx);
    if (rc) return rc;
  }
  if (gm) {
    ctx->gm_cx = job.carry.cx;
    ctx->gm_cy = job.carry.cy;
    ctx->gm_prob = job.carry.prob;
  }
  job.delta(out_delta);
  *out_prob = job.best_prob;
  if (m->has_obs && m->obs.on_matching_end) m->obs.on_matching_end(m->obs.user, out_delta, job.best_prob);
  return SLAMHIP_OK;
}

}  // extern "C"
