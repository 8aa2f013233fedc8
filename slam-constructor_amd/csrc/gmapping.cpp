// gmapping.cpp -- GMapping particle-filter step over the GPU scorer.
//
// Reference behaviour restated (paths relative to the reference root):
//   GmappingWorld                 src/slams/gmapping/gmapping_world.h:36-127
//     update_robot_pose :57-71, handle_observation :73-101, mark_master :103-110,
//     reset_scan_matching_delta :116-119
//   GmappingParticleFilter        src/slams/gmapping/gmapping_particle_filter.h:29-118
//     handle_sensor_data :45-50, update_robot_pose :52-57, handle_observation :70-77,
//     try_resample :88-99 (fabs of a bool, Q23), ensure_master_exists :102-113
//   ParticleFilter / UniformResamling  src/core/particle_filter.h:34-66,70-121
//   init_gmapping                 src/slams/gmapping/init_gmapping.h:49-65 (HC(6, 0.1, 0.1), WMPP with
//                                 even weights, GmappingOccupancyObservationPE(0.1, 1))
//
// What runs where: the per-particle state machines (gate, pose noise, RNG streams, weights,
// N_eff, resampling, master hand-over) stay on the host, bit for bit; every particle that takes the
// scan-matching branch owns a MatchJob (matchers.h) and ALL jobs advance in lock-step through
// shared launches of the GMapping kernel (K3).  Particles are independent between the odometry
// update and the weight normalisation, so a filter object may hold just a shard [first,
// first+count) of the particles: the caller all-gathers the raw weights (and, when a resampling
// happens, the particle records) between predict_match and plan_resample/import.
//
// The reference shares ONE OOPE cache among all particles (Q20) and evaluates them one after
// another; in lock-step a particle's predecessor is not finished when it starts, so every job
// starts without a carry-in and the chain is verified afterwards: a particle whose first run would
// have hit its predecessor's final cache entry is re-matched alone with that carry (counted in
// `carry_reruns`; rare: it needs the same endpoint cell AND a different cached value).
//
// Map update (append_scan, gmapping_world.h:93-97): off by default -- the lock-step step equals the
// reference run with slam/mapping/max_range = 0.  slamhip_gmapping_set_map_update switches to the
// reference's full semantics: the particles share ONE map (Q20) and each appends its scan before
// the next one matches, so the step runs particle after particle (GPU match, then K6 on the same
// HBM map); that mode cannot be sharded.

#include <algorithm>
#include <cstdlib>
#include <string>
#include <type_traits>

#include "matchers.h"
#include "tile_pool.h"

namespace slamhip {

struct GmParticle {
  double pose[3], raw_odom[3], weight;
  int is_master, scan_is_first;
  std::mt19937 eng;
  NormalRV guess[3];  // _pose_guess_rv
  int nsd_is_normal;  // _next_sm_delta_rv: UniformRV1D(a, b) or, for a master, GaussianRV1D(0, 0)
  double nsd_a[3], nsd_b[3];
  NormalRV nsd_norm[3];
  double dsl[3], nsd[3];  // _delta_since_last_sm, _next_sm_delta
};
static_assert(std::is_trivially_copyable<GmParticle>::value, "particle records travel as raw bytes");

static void reset_sm_delta(GmParticle &p) {
  p.dsl[0] = p.dsl[1] = p.dsl[2] = 0;
  for (int k = 0; k < 3; ++k)
    p.nsd[k] = p.nsd_is_normal ? p.nsd_norm[k](p.eng) : canonical(p.eng) * (p.nsd_b[k] - p.nsd_a[k]) + p.nsd_a[k];
}

static void init_particle(GmParticle &p, const slamhip_gmapping_params &gp, uint32_t seed, double weight) {
  std::memset(static_cast<void *>(&p), 0, sizeof(p));
  p.weight = weight;
  p.scan_is_first = 1;
  p.eng = std::mt19937(seed);
  p.guess[0] = NormalRV(gp.mean_sample_xy, gp.sigma_sample_xy);
  p.guess[1] = NormalRV(gp.mean_sample_xy, gp.sigma_sample_xy);
  p.guess[2] = NormalRV(gp.mean_sample_th, gp.sigma_sample_th);
  p.nsd_is_normal = 0;
  p.nsd_a[0] = p.nsd_a[1] = gp.min_sm_lim_xy;
  p.nsd_b[0] = p.nsd_b[1] = gp.max_sm_lim_xy;
  p.nsd_a[2] = gp.min_sm_lim_th;
  p.nsd_b[2] = gp.max_sm_lim_th;
  for (int k = 0; k < 3; ++k) p.nsd_norm[k] = NormalRV(0, 0);
  reset_sm_delta(p);
}

static void mark_master(GmParticle &p) {
  p.is_master = 1;
  for (int k = 0; k < 3; ++k) {
    p.guess[k] = NormalRV(0, 0);
    p.nsd_norm[k] = NormalRV(0, 0);
  }
  p.nsd_is_normal = 1;
}

// `*new_particle = *sampled; new_particle->sample()` (particle_filter.h:94-96): everything is
// copied, the RobotPoseDeltaRV members through clone(), i.e. fresh distributions with the same
// parameters (robot_pose.h:72-81), and the master flag is cleared
static void copy_as_duplicate(GmParticle &dst, const GmParticle &src) {
  dst = src;
  for (int k = 0; k < 3; ++k) {
    dst.guess[k] = NormalRV(src.guess[k].mean, src.guess[k].stddev);
    dst.nsd_norm[k] = NormalRV(src.nsd_norm[k].mean, src.nsd_norm[k].stddev);
  }
  dst.is_master = 0;
}

}  // namespace slamhip

using namespace slamhip;

struct slamhip_gmapping {
  slamhip_ctx *ctx = nullptr;
  slamhip_gmapping_params prm{};
  slamhip_spe_cfg cfg{};
  int n_total = 0, first = 0, count = 0;
  std::vector<GmParticle> p;  // local shard
  double traversed[3] = {0, 0, 0};
  GmCarry carry;  // the shared OOPE cache as this shard sees it
  slamhip_matcher *sm = nullptr;  // lone matches of the shared-map mode
  GmMultiChain *mc = nullptr;     // likelihood-only steps on a dense map: one device chain per particle
  std::vector<GmChainResult> chain_out;
  std::vector<double> chain_inits;
  GmCarry step_carry;  // sharded steps: the cache entry the previous step ended with (on every shard)
  std::vector<MatchJob> jobs;
  std::vector<HillClimbingPoseEnumerator> pes;
  std::vector<double> all_w;  // normalised weights of all particles (after plan_resample)
  // sin / cos of the raw beam angles: a scanner's angles do not change from scan to scan, so they are
  // recomputed only when the angle array does (1080 sincos calls per step were 30 us of a 0.85 ms step)
  std::vector<double> trig_angle, trig_cos, trig_sin;
  long long scorer_calls = 0, poses_evaluated = 0, launches = 0, carry_reruns = 0;
  // map update inside the step (gmapping_world.h:93-97): sequential, unsharded filters only
  bool update = false;
  slamhip_scan_adder_cfg upd{};
  long long cell_updates = 0;
  // per-particle copy-on-write maps: particle i owns slot i of the pool; matching stays
  // in lock-step and the map updates of all matched particles run as ONE batched K6
  TilePool *tp = nullptr;
  TiledTarget tt{};
  bool maps_handled_by_caller = false;  // set by slamhip_gmapping_import_maps around the particle import
  // a step in three phases (slamhip_gmapping_match_begin / _carry_fix / _match_finish): what the first
  // leaves for the others
  std::vector<int> act_idx;
  std::vector<MatchJob *> act;
  bool pending = false;      // begin ran, finish did not yet
  // sharded steps: a failure behind the step's last collective travels in the status word of the next step's first
  // one, so that every rank leaves that step together instead of waiting for a rank that has given up
  int deferred_rc = 0;
  // testing (slamhip_gmapping_debug_fail): make the n-th call from now of one place fail as a rank-local error --
  // 1 match_finish, 2 the export of migrating maps (between the migration's collectives), 3 the final map import
  int debug_fail_where = 0, debug_fail_countdown = 0;
  // device staging of the tile contents of migrating maps (slamhip_gmapping_step_sharded with per-particle maps)
  char *d_mig_send = nullptr, *d_mig_recv = nullptr;
  size_t mig_send_cap = 0, mig_recv_cap = 0;
  long long maps_migrated = 0, map_bytes_sent = 0;
  bool chained = false;      // the jobs of this step ran in lock-step (the carry chain can be re-checked)
  bool shard_chain = false;  // sharded step: the first job starts WITHOUT a carry and is checked afterwards
                             // against the predecessor shard's final cache entry (carry_fix)
  int pending_map = 0;
  std::vector<int> shard_counts;   // particles per rank (slamhip_gmapping_step_sharded)
  std::vector<double> scan_range;  // the raw scan of the step in flight (finish appends it to the maps)
  std::vector<int> scan_occ;
  bool scan_has_occ = false;
};

namespace {

int bad(const char *msg) {
  set_error(msg);
  return SLAMHIP_ERR_INVALID;
}

// the testing hook's trigger: true when place `where` is to fail now
bool debug_fail_now(slamhip_gmapping *g, int where) {
#ifndef SLAMHIP_TESTING
  (void)g;
  (void)where;
  return false;  // (no injected failures in the shipped library)
#endif
  if (g->debug_fail_where != where) return false;
  if (--g->debug_fail_countdown > 0) return false;
  g->debug_fail_where = 0;
  set_error("injected failure (slamhip_gmapping_debug_fail)");
  return true;
}

int heaviest(const std::vector<double> &w) {
  int h = -1;
  for (int i = 0; i < (int)w.size(); ++i) {
    if (h >= 0 && w[i] < w[h]) continue;
    h = i;
  }
  return h;
}

// drives a set of jobs to completion through shared launches
// `slots` (per-particle maps only): the map slot of every job
// sin / cos per raw beam, cached over steps while the angles stay the same
void raw_trig(slamhip_gmapping *g, int n_raw, const double *angle) {
  if ((int)g->trig_angle.size() == n_raw && std::memcmp(g->trig_angle.data(), angle, sizeof(double) * n_raw) == 0)
    return;
  g->trig_angle.assign(angle, angle + n_raw);
  g->trig_cos.resize(n_raw);
  g->trig_sin.resize(n_raw);
  slamhip_beam_trig_raw(n_raw, angle, g->trig_cos.data(), g->trig_sin.data());
}

// the scoring view of the filter's per-particle maps as they are now
void bind_tiled_target(slamhip_gmapping *g) {
  TiledTarget &t = g->tt;
  TilePool *tp = g->tp;
  // (the tiles' neighbourhood masks: derived once, kept by the writers; a failure here leaves the nine-cell scorer)
  if (g->cfg.gm_window == 1 && tile_pool_nbr_masks(tp, g->cfg.gm_fullness_th) != SLAMHIP_OK) tp->nbr_ok = false;
  t.nbr_ok = (tp->nbr_ok && g->cfg.gm_window == 1 && tp->nbr_th == g->cfg.gm_fullness_th) ? 1 : 0;
  t.nbr_th = tp->nbr_th;
  t.pool = tp->d_pool;
  t.tables = tp->d_table();
  t.table_stride = tp->table_stride();
  t.tiles_x = tp->tiles_x;
  t.width = tp->width();
  t.height = tp->height();
  t.origin_x = tp->origin_x;
  t.origin_y = tp->origin_y;
  t.scale = tp->scale;
  for (int k = 0; k < 4; ++k) t.unknown[k] = tp->unknown[k];
}

int run_jobs(slamhip_gmapping *g, int map_id, std::vector<MatchJob *> &act, int per_job_budget,
             const int *slots = nullptr) {
  slamhip_ctx *ctx = g->ctx;
  const int n_jobs = (int)act.size();
  std::vector<int> off(n_jobs), cnt(n_jobs);
  int rc = ensure_pose_capacity(ctx, (per_job_budget + 1) * n_jobs);
  if (rc) return rc;
  if (g->tp) bind_tiled_target(g);  // the table buffer flips on resampling, the extent grows with the maps
  // Groups of jobs take turns (SLAMHIP_PF_PIPELINE = number of groups, default 2; 1 = one launch per
  // round): while the GPU scores one group's batch the host replays and re-plans another.  Each group
  // owns one window of the staging buffers; jobs are independent, so the interleaving changes no
  // result.  Measured at 100 particles: with the first K3 (40 us per full launch, two half launches
  // 2 x 33 us) the pipeline gained 5 % and stayed off; with the present K3 (25 us per full launch against
  // ~35 us of host work per round) it hides most of the host: 1.24 -> 0.98 ms per step.
  constexpr int groups_env = 2;  // (1 = one launch per round, 3 and 4 measured slower: the launches get too small)
  constexpr int kMaxGroups = 4;
  const int n_groups = (ctx->low_latency && !ctx->stage_poses && n_jobs >= 16)
                           ? std::max(1, std::min(kMaxGroups, groups_env)) : 1;
  struct Group {
    int lo, hi, base, total, lane;
    unsigned seq;
    bool in_flight;
  } grp[kMaxGroups];
  // odd groups launch on the context's second stream: their kernel starts while the even group's is
  // still draining and publishing (SLAMHIP_PF_LANES=1: everything on one stream)
  constexpr bool two_lanes = true;  // (both groups on one stream: 0.97 against 0.94 ms per step)
  if (n_groups > 1 && two_lanes) {
    rc = lane_fork(ctx);
    if (rc) return rc;
  }
  for (int q = 0; q < n_groups; ++q) {
    grp[q].lo = q * n_jobs / n_groups;
    grp[q].hi = (q + 1) * n_jobs / n_groups;
    grp[q].base = grp[q].lo * (per_job_budget + 1);
    grp[q].in_flight = false;
    grp[q].lane = (n_groups > 1 && two_lanes) ? (q & 1) : 0;
  }
  auto plan_and_submit = [&](Group &G) -> int {
    int total = 0;
    for (int k = G.lo; k < G.hi; ++k) {
      off[k] = G.base + total;
      cnt[k] = act[k]->done ? 0 : act[k]->plan(per_job_budget, ctx->h_poses + 3 * (size_t)off[k]);
      if (slots)
        for (int q = 0; q < cnt[k]; ++q) ctx->h_pose_slot[off[k] + q] = slots[k];
      total += cnt[k];
    }
    G.total = total;
    G.in_flight = total > 0;
    if (!total) return SLAMHIP_OK;
    g->launches += 1;
    g->poses_evaluated += total;
    return score_staged(ctx, map_id, &g->cfg, total, slots ? &g->tt : nullptr, G.base, &G.seq, G.lane);
  };
  for (int q = 0; q < n_groups; ++q) {
    rc = plan_and_submit(grp[q]);
    if (rc) return rc;
  }
  auto any_in_flight = [&]() {
    for (int q = 0; q < n_groups; ++q)
      if (grp[q].in_flight) return true;
    return false;
  };
  while (any_in_flight()) {
    for (int q = 0; q < n_groups; ++q) {
      Group &G = grp[q];
      if (!G.in_flight) continue;
      rc = score_wait(ctx, G.seq, G.lane);
      if (rc) return rc;
      for (int k = G.lo; k < G.hi; ++k) {
        if (cnt[k] == 0) continue;
        rc = act[k]->consume(ctx->h_scores + off[k], ctx->h_gm_info + off[k], ctx);
        if (rc) return rc;
      }
      rc = plan_and_submit(G);
      if (rc) return rc;
    }
  }
  return SLAMHIP_OK;
}

}  // namespace

extern "C" {

size_t slamhip_gmapping_blob_size(void) { return sizeof(GmParticle); }

int slamhip_gmapping_create(slamhip_ctx *ctx, const slamhip_gmapping_params *prm, int n_total, int first,
                            int count, const uint32_t *seeds, slamhip_gmapping **out) {
  // ctx may be null for host-only use (weights / resampling bookkeeping of a shard); matching then
  // fails loudly
  if (!prm || !seeds || !out) return bad("null argument");
  if (n_total <= 0 || first < 0 || count <= 0 || first + count > n_total) return bad("bad particle shard");
  auto *g = new slamhip_gmapping;
  g->ctx = ctx;
  g->prm = *prm;
  std::memset(&g->cfg, 0, sizeof(g->cfg));
  g->cfg.oope = SLAMHIP_OOPE_GMAPPING;
  g->cfg.oie = SLAMHIP_OIE_DISCREPANCY;
  g->cfg.gm_fullness_th = prm->oope_fullness_th;
  g->cfg.gm_window = prm->oope_window;
  g->cfg.sum_order = SLAMHIP_SUM_TREE256;
  g->cfg.pose_trig = prm->pose_trig;
  g->n_total = n_total;
  g->first = first;
  g->count = count;
  g->p.resize(count);
  for (int i = 0; i < count; ++i) init_particle(g->p[i], *prm, seeds[i], 1.0 / n_total);
  // the heaviest particle becomes the master; with equal weights that is the LAST one
  // (particle_filter.h:114-121, gmapping_particle_filter.h:39-42)
  if (first + count == n_total) mark_master(g->p[count - 1]);
  g->jobs.resize(count);
  g->all_w.assign(n_total, 1.0 / n_total);
  *out = g;
  return SLAMHIP_OK;
}

int slamhip_gmapping_destroy(slamhip_gmapping *g) {
  if (g && g->sm) slamhip_matcher_destroy(g->sm);
  if (g && g->mc) gm_multi_chain_free(g->mc);
  if (g && g->tp) tile_pool_destroy(g->tp);
  if (g && g->d_mig_send) hipFree(g->d_mig_send);
  if (g && g->d_mig_recv) hipFree(g->d_mig_recv);
  delete g;
  return SLAMHIP_OK;
}

// the shared-cache chain in the reference's particle order, from job `from` on: a particle whose first
// run would have hit its predecessor's final cache entry with another value is re-matched alone with
// that carry
// (from == 0: job 0 started without a carry as well and is checked against `before`, the cache entry the step
// starts from)
static int verify_chain(slamhip_gmapping *g, int map_id, size_t from, const GmCarry *before = nullptr) {
  std::vector<MatchJob *> &act = g->act;
  const std::vector<int> &act_idx = g->act_idx;
  if (act.empty()) return SLAMHIP_OK;
  int rc;
  GmCarry prev = from > 0 ? act[from - 1]->carry : (before ? *before : GmCarry{});
  for (size_t k = from; k < act.size(); ++k) {
    MatchJob &job = *act[k];
    const GmPoseInfo &fi = job.first_info;
    if (prev.prob != -1.0 && fi.first_cx == prev.cx && fi.first_cy == prev.cy && prev.prob != fi.v0) {
      g->pes[k] = HillClimbingPoseEnumerator(g->prm.hc_failed_rounds_limit, g->prm.hc_translation,
                                             g->prm.hc_rotation);
      const GmParticle &p = g->p[act_idx[k]];
      job.start(&g->pes[k], Pose{p.pose[0], p.pose[1], p.pose[2]}, true, nullptr, prev, 0.25);
      std::vector<MatchJob *> one{&job};
      rc = run_jobs(g, map_id, one, 126, g->tp ? &act_idx[k] : nullptr);
      if (rc) return rc;
      g->carry_reruns += 1;
    }
    prev = job.carry;
  }
  g->carry = prev;
  return SLAMHIP_OK;
}

int slamhip_gmapping_match_begin(slamhip_gmapping *g, int map_id, int n_raw, const double *range,
                                 const double *angle, const int *is_occ, const double odom_delta[3]) {
  if (!g || !range || !angle || !odom_delta) return bad("null argument");
  if (g->pending) return bad("the previous step was not finished (slamhip_gmapping_match_finish)");
  slamhip_ctx *ctx = g->ctx;
  if (!ctx) {
    set_error("this filter was created without a GPU context; there is no CPU scorer");
    return SLAMHIP_ERR_NO_DEVICE;
  }
  SLAMHIP_CHECK(hipSetDevice(ctx->device));
  g->scorer_calls = g->poses_evaluated = g->launches = g->carry_reruns = g->cell_updates = 0;
  const double *d = odom_delta;
  // update_robot_pose: the odometry delta is rotated by the particle's accumulated heading correction
  for (auto &p : g->p) {
    double s, c;
    ::sincos(p.pose[2] - p.raw_odom[2], &s, &c);
    const double cx = c * d[0] - s * d[1], cy = s * d[0] + c * d[1], ct = d[2];
    for (int k = 0; k < 3; ++k) p.raw_odom[k] += d[k];
    p.dsl[0] += std::fabs(cx);
    p.dsl[1] += std::fabs(cy);
    p.dsl[2] += std::fabs(ct);
    p.pose[0] += cx;
    p.pose[1] += cy;
    p.pose[2] += ct;
  }
  for (int k = 0; k < 3; ++k) g->traversed[k] += std::fabs(d[k]);

  // gate + pose noise, in particle order
  std::vector<int> &act_idx = g->act_idx;
  act_idx.clear();
  g->act.clear();
  g->chained = false;
  g->pending = true;
  g->pending_map = map_id;
  g->scan_range.assign(range, range + n_raw);
  g->scan_has_occ = is_occ != nullptr;
  if (is_occ) g->scan_occ.assign(is_occ, is_occ + n_raw);
  for (int i = 0; i < g->count; ++i) {
    GmParticle &p = g->p[i];
    if (p.dsl[0] * p.dsl[0] + p.dsl[1] * p.dsl[1] < p.nsd[0] * p.nsd[0] + p.nsd[1] * p.nsd[1] &&
        std::fabs(p.dsl[2]) < p.nsd[2])
      continue;
    if (!p.scan_is_first) {
      const double nx = p.guess[0](p.eng);
      const double ny = p.guess[1](p.eng);
      const double nt = p.guess[2](p.eng);
      p.pose[0] += nx;
      p.pose[1] += ny;
      p.pose[2] += nt;
    }
    act_idx.push_back(i);
  }
  if (!act_idx.empty()) {
    // filter_scan: GMapping maps are unbounded (has_cell is always true, lazy_tiled_grid_map.h:150),
    // so the filtered scan does not depend on the particle
    std::vector<int> kept(n_raw > 0 ? n_raw : 1);
    int nk = 0;
    int rc = slamhip_filter_scan(n_raw, range, angle, is_occ, SLAMHIP_TRIG_RAW, 0, 1, 0, nullptr, nullptr,
                                 g->p[act_idx[0]].pose, g->prm.sp_skip_rate, g->prm.sp_max_usable_range,
                                 /*bounded*/ 0, 0, 0, 0, 0, 1.0, kept.data(), &nk);
    if (rc) return rc;
    if (nk <= 0) return bad("no usable scan points");
    raw_trig(g, n_raw, angle);
    std::vector<double> fr(nk), fa(nk), fw(nk), fc(nk), fs(nk);
    for (int k = 0; k < nk; ++k) {
      fr[k] = range[kept[k]];
      fa[k] = angle[kept[k]];
      fc[k] = g->trig_cos[kept[k]];
      fs[k] = g->trig_sin[kept[k]];
    }
    slamhip_scan_weights(0, nk, fr.data(), fa.data(), fw.data());
    rc = slamhip_scan_upload(ctx, nk, fr.data(), fc.data(), fs.data(), fw.data(), nullptr);
    if (rc) return rc;
    // SLAMHIP_POSE_TRIG_RAW_EXACT (the reference's default provider bit for bit, exact_kernels.hip): the scorer adds the
    // pose heading to the kept points' angles itself, and the reference's ONE cache object lives on the device and is
    // applied in call order -- particle after particle, candidate after candidate: the sequential loop below, whether
    // or not the map is updated
    const bool exact = g->cfg.pose_trig == SLAMHIP_POSE_TRIG_RAW_EXACT;
    if (exact) {
      if (g->tp) return bad("SLAMHIP_POSE_TRIG_RAW_EXACT with per-particle maps is not built: use the shared map");
      rc = slamhip_scan_set_angles(ctx, nk, fa.data());
      if (rc) return rc;
    }

    if (g->update || exact) {
      // The reference's full step: every matching particle appends its scan to the ONE shared map
      // before the next particle matches (gmapping_world.h:88-99, Q20), so the particles are
      // strictly sequential: match on the GPU (lone-matcher speculation), then K6 on the same map.
      const std::vector<double> &rc_all = g->trig_cos, &rs_all = g->trig_sin;  // raw_trig() above
      g->pes.clear();
      g->pes.emplace_back(g->prm.hc_failed_rounds_limit, g->prm.hc_translation, g->prm.hc_rotation);
      // ... and its completion is not awaited: the next particle's chain is queued behind the update on the same
      // stream, the status words are collected when the loop is through
      struct ReuseGuard {
        slamhip_ctx *c;
        bool was_deferred;  // (slamhip_map_set_deferred: the caller's own setting comes back afterwards)
        explicit ReuseGuard(slamhip_ctx *cc) : c(cc) {
          mu_allow_scan_reuse(c, true);
          was_deferred = mu_set_deferred(c, true);
        }
        ~ReuseGuard() {
          mu_drain(c, nullptr, nullptr);  // (an early return: nothing may stay in flight)
          mu_set_deferred(c, was_deferred);
          mu_allow_scan_reuse(c, false);
        }
      } reuse_guard(ctx);
      // one matcher object for the lone matches: with device pose trig its accept chain runs on the device
      // (hc_chain.hip), otherwise through host-driven batches -- the same scorer calls either way
      if (!g->sm) {
        rc = slamhip_matcher_create_hc(ctx, &g->cfg, g->prm.hc_failed_rounds_limit, g->prm.hc_translation,
                                       g->prm.hc_rotation, &g->sm);
        if (rc) return rc;
      }
      // (a lone GMapping chain stays a chain of kernels: the co-resident form is no faster for ONE chain, see
      // resident_wanted in matchers.cpp)
      for (int idx : act_idx) {
        GmParticle &p = g->p[idx];
        // the filter's cache is the context's for the duration of the match
        ctx->gm_cx = g->carry.cx;
        ctx->gm_cy = g->carry.cy;
        ctx->gm_prob = g->carry.prob;
        double dl[3], best_prob = 0.0;
        rc = slamhip_matcher_process_scan(g->sm, map_id, p.pose, dl, &best_prob);
        if (rc) return rc;
        g->carry = GmCarry{ctx->gm_cx, ctx->gm_cy, ctx->gm_prob};
        long long calls = 0, evaluated = 0, launches = 0;
        slamhip_matcher_stats(g->sm, &calls, &evaluated, &launches);
        g->poses_evaluated += evaluated;
        g->launches += launches;
        for (int c = 0; c < 3; ++c) p.pose[c] += dl[c];
        if (!g->update) {
          p.scan_is_first = 0;
        } else if (0.0 < best_prob || p.scan_is_first) {
          slamhip_scan_adder_cfg cfg = g->upd;
          cfg.rule = SLAMHIP_RULE_GMAPPING;
          cfg.scan_quality = 1.0;  // scan.quality handed to append_scan (gmapping_world.h:95)
          long long nu = 0;
          rc = exact ? slamhip_map_append_scan_raw(ctx, map_id, &cfg, p.pose, n_raw, range, angle, is_occ, nullptr, &nu)
                     : slamhip_map_append_scan(ctx, map_id, &cfg, p.pose, n_raw, range, rc_all.data(), rs_all.data(),
                                               is_occ, &nu);
          if (rc) return rc;
          if (nu >= 0) g->cell_updates += nu;  // (deferred: counted by mu_drain below)
          p.scan_is_first = 0;
        }
        p.weight = best_prob * p.weight;
        reset_sm_delta(p);
        g->scorer_calls += calls;
      }
      {
        long long nu = 0;
        int uerr = 0;
        rc = mu_drain(ctx, &nu, &uerr);
        if (rc) return rc;
        g->cell_updates += nu;
        if (uerr) {
          set_error(uerr == 2 ? "internal: the device counted more cell updates than the host sized the buffers for"
                              : "a beam leaves the bound map window: bind a larger map before the filter runs");
          return SLAMHIP_ERR_STATE;
        }
      }
      act_idx.clear();  // everything is applied already: nothing left for match_finish
      return SLAMHIP_OK;
    }
    g->pes.clear();
    g->pes.reserve(act_idx.size());
    std::vector<MatchJob *> &act = g->act;
    for (size_t k = 0; k < act_idx.size(); ++k) {
      g->pes.emplace_back(g->prm.hc_failed_rounds_limit, g->prm.hc_translation, g->prm.hc_rotation);
      GmParticle &p = g->p[act_idx[k]];
      MatchJob &job = g->jobs[act_idx[k]];
      job.start(&g->pes[k], Pose{p.pose[0], p.pose[1], p.pose[2]}, true, nullptr,
                (k == 0 && !g->shard_chain) ? g->carry : GmCarry{}, 0.25);
      act.push_back(&job);
    }
    // Device pose trig, 3x3 window: every particle's accept chain runs on the device, all chains
    // in shared launches (hc_chain.hip, grid.y = particle); no chain starts from a cache entry, the hand-overs --
    // the step's own included -- are checked afterwards in particle order like the lock-step jobs'.
    const bool pf_chain_off = !ctx->filter_chains;  // (SLAMHIP_OPT_FILTER_CHAINS)
    // (per-particle maps: every chain gathers through its particle's tile table; measured, ms per step, chains /
    // lock-step: 13 particles 0.74 / 0.89, 100 particles as a chain of kernels 1.66 / 1.64 -- so for shards of up to
    // 64 particles, and -- r04 -- for as many as fit ONE co-resident launch: 100 particles 1.53 / 1.68; cfg5's 500
    // do not fit, and in groups of 100 or 64 per launch they are slower than lock-step, 10.8 and 11.6 / 9.1 ms)
    const bool chains = !pf_chain_off && (!g->tp || act.size() <= 64 || gm_multi_chain_fits_resident(ctx, (int)act.size())) && g->cfg.pose_trig == SLAMHIP_POSE_TRIG_DEVICE && g->cfg.gm_window == 1 &&
                        g->cfg.sum_order == SLAMHIP_SUM_TREE256 && ctx->scan_n <= 1280 && ctx->low_latency &&
                        !ctx->stage_poses && g->prm.hc_failed_rounds_limit >= 1 && g->prm.hc_failed_rounds_limit <= 250;
    if (chains) {
      const int na = (int)act.size();
      g->chain_inits.resize(3 * (size_t)na);
      g->chain_out.resize(na);
      for (int k = 0; k < na; ++k) {
        const GmParticle &p = g->p[act_idx[k]];
        for (int c = 0; c < 3; ++c) g->chain_inits[3 * k + c] = p.pose[c];
      }
      long long kernels = 0;
      if (g->tp) bind_tiled_target(g);  // (slot = particle index)
      rc = gm_multi_chain_run(ctx, &g->mc, map_id, &g->cfg, g->prm.hc_failed_rounds_limit, g->prm.hc_translation,
                              g->prm.hc_rotation, na, g->chain_inits.data(), g->chain_out.data(), &kernels,
                              g->tp ? &g->tt : nullptr, g->tp ? act_idx.data() : nullptr);
      if (rc) return rc;
      g->launches += kernels;
      const GmCarry before = g->shard_chain ? GmCarry{} : g->carry;
      for (int k = 0; k < na; ++k) {
        MatchJob &job = *act[k];
        const GmChainResult &r = g->chain_out[k];
        if (r.error == 3) {
          // a scan that is one run sat on this chain's path: its cache hand-over needs the sequential replay
          std::vector<MatchJob *> one{&job};  // (started above, without a carry)
          job.carry = GmCarry{};
          job.carry_in = GmCarry{};
          rc = run_jobs(g, map_id, one, 126, g->tp ? &act_idx[k] : nullptr);
          if (rc) return rc;
          continue;
        }
        job.best = Pose{r.pose[0], r.pose[1], r.pose[2]};
        job.best_prob = r.prob;
        job.carry = GmCarry{r.cx, r.cy, r.cprob};
        job.carry_in = GmCarry{};
        job.first_info = r.first_info;
        job.first_raw_score = r.first_raw;
        job.scorer_calls = r.calls;
        job.poses_evaluated = r.evaluated;
        job.launches = r.steps;
        job.first = false;
        job.done = true;
        g->poses_evaluated += r.evaluated;
      }
      g->chained = true;
      rc = verify_chain(g, map_id, 0, &before);
      if (rc) return rc;
      return SLAMHIP_OK;
    }
    int per_job = std::max(6, std::min(126, 12288 / (int)act.size() / 6 * 6));
    double min_reach = 0.3;  // measured on MI355X, 100 particles: 1.9 ms/step at 0.3 vs 6.7 ms at 0.01
    for (MatchJob *j : act) {
      j->tree.min_reach = min_reach;
      j->timed = false;
    }
    rc = run_jobs(g, map_id, act, per_job, g->tp ? act_idx.data() : nullptr);  // slot = particle index
    if (rc) return rc;
    g->chained = true;
    rc = verify_chain(g, map_id, 1);
    if (rc) return rc;
  }
  return SLAMHIP_OK;
}

int slamhip_gmapping_match_finish(slamhip_gmapping *g, double *raw_weights_out) {
  if (!g) return bad("null filter");
  if (!g->pending) return bad("no step in flight (slamhip_gmapping_match_begin)");
  slamhip_ctx *ctx = g->ctx;
  SLAMHIP_CHECK(hipSetDevice(ctx->device));
  g->pending = false;
  if (debug_fail_now(g, 1)) return SLAMHIP_ERR_STATE;
  const std::vector<int> &act_idx = g->act_idx;
  std::vector<MatchJob *> &act = g->act;
  const int n_raw = (int)g->scan_range.size();
  const double *range = g->scan_range.data();
  const int *is_occ = g->scan_has_occ ? g->scan_occ.data() : nullptr;
  int rc;
  if (!act.empty()) {
    std::vector<double> upd_pose;
    std::vector<int> upd_slot;
    for (size_t k = 0; k < act_idx.size(); ++k) {
      GmParticle &p = g->p[act_idx[k]];
      const MatchJob &job = *act[k];
      double dl[3];
      job.delta(dl);
      for (int c = 0; c < 3; ++c) p.pose[c] += dl[c];
      if (0.0 < job.best_prob || p.scan_is_first) {
        // gmapping_world.h:93-97: the particle appends the scan to ITS map from the corrected pose
        if (g->tp) {
          upd_pose.insert(upd_pose.end(), p.pose, p.pose + 3);
          upd_slot.push_back(act_idx[k]);
        }
        p.scan_is_first = 0;
      }
      p.weight = job.best_prob * p.weight;
      reset_sm_delta(p);
      g->scorer_calls += job.scorer_calls;
    }
    if (g->tp && !upd_slot.empty()) {
      // own maps: no particle reads another's update, so all appends of the step form one batch
      const std::vector<double> &rc_all = g->trig_cos, &rs_all = g->trig_sin;  // raw_trig() above
      slamhip_scan_adder_cfg cfg = g->upd;
      cfg.rule = SLAMHIP_RULE_GMAPPING;
      cfg.scan_quality = 1.0;
      long long nu = 0;
      rc = mu_append_batch(ctx, g->tp, &cfg, (int)upd_slot.size(), upd_pose.data(), upd_slot.data(), n_raw, range,
                           rc_all.data(), rs_all.data(), is_occ, &nu);
      if (rc) return rc;
      g->cell_updates += nu;
    }
  }
  if (raw_weights_out)
    for (int i = 0; i < g->count; ++i) raw_weights_out[i] = g->p[i].weight;
  return SLAMHIP_OK;
}

int slamhip_gmapping_predict_match(slamhip_gmapping *g, int map_id, int n_raw, const double *range,
                                   const double *angle, const int *is_occ, const double odom_delta[3],
                                   double *raw_weights_out) {
  if (!g) return bad("null filter");
  int rc = slamhip_gmapping_match_begin(g, map_id, n_raw, range, angle, is_occ, odom_delta);
  if (rc) {
    g->pending = false;
    return rc;
  }
  return slamhip_gmapping_match_finish(g, raw_weights_out);
}

// ---- the shared OOPE cache across shards (Q20) ------------------------------------------------------
// In the reference ONE cache object is handed from particle to particle; a shard's first matching
// particle therefore continues the cache of the last matching particle BEFORE it -- on the previous
// shard, or, for the first one of a step, the last one of the previous step.  Sharded steps start every
// shard's first job without a carry; afterwards the shards exchange these records and each one checks its
// first job against its predecessor's final entry (re-matching on a hit, like verify_chain does inside a
// shard) until nothing changes any more.
int slamhip_gmapping_carry_record(slamhip_gmapping *g, slamhip_carry_record *rec) {
  if (!g || !rec) return bad("null argument");
  std::memset(rec, 0, sizeof(*rec));
  rec->has_active = (g->pending && g->chained && !g->act.empty()) ? 1 : 0;
  if (rec->has_active) {
    const GmPoseInfo &fi = g->act[0]->first_info;
    rec->first_cx = fi.first_cx;
    rec->first_cy = fi.first_cy;
    rec->first_v0 = fi.v0;
  }
  rec->carry_cx = g->carry.cx;
  rec->carry_cy = g->carry.cy;
  rec->carry_prob = g->carry.prob;
  return SLAMHIP_OK;
}

int slamhip_gmapping_carry_fix(slamhip_gmapping *g, const slamhip_carry_record *all, int world, int rank,
                               int *changed) {
  if (!g || !all || !changed || rank < 0 || rank >= world) return bad("bad argument");
  *changed = 0;
  // the cache as it reaches this shard: the final entry of the nearest earlier shard that matched, or
  // what the previous step left (every shard keeps that in `step_carry`)
  GmCarry pred = g->step_carry;
  for (int r = 0; r < rank; ++r)
    if (all[r].has_active) pred = GmCarry{all[r].carry_cx, all[r].carry_cy, all[r].carry_prob};
  if (!(g->pending && g->chained) || g->act.empty()) {
    g->carry = pred;  // nothing matched here: the cache passes through
    return SLAMHIP_OK;
  }
  MatchJob &job = *g->act[0];
  const GmPoseInfo &fi = job.first_info;
  const bool same_in = job.carry_in.prob == pred.prob && job.carry_in.cx == pred.cx && job.carry_in.cy == pred.cy;
  const bool hit = pred.prob != -1.0 && fi.first_cx == pred.cx && fi.first_cy == pred.cy && pred.prob != fi.v0;
  const bool had_hit = job.carry_in.prob != -1.0 && fi.first_cx == job.carry_in.cx && fi.first_cy == job.carry_in.cy &&
                       job.carry_in.prob != fi.v0;
  if (same_in || (!hit && !had_hit)) return SLAMHIP_OK;  // the job ran with an equivalent cache
  slamhip_ctx *ctx = g->ctx;
  SLAMHIP_CHECK(hipSetDevice(ctx->device));
  const GmCarry before = g->carry;
  g->pes[0] = HillClimbingPoseEnumerator(g->prm.hc_failed_rounds_limit, g->prm.hc_translation, g->prm.hc_rotation);
  const GmParticle &p = g->p[g->act_idx[0]];
  job.start(&g->pes[0], Pose{p.pose[0], p.pose[1], p.pose[2]}, true, nullptr, pred, 0.25);
  std::vector<MatchJob *> one{&job};
  int rc = run_jobs(g, g->pending_map, one, 126, g->tp ? &g->act_idx[0] : nullptr);
  if (rc) return rc;
  g->carry_reruns += 1;
  rc = verify_chain(g, g->pending_map, 1);
  if (rc) return rc;
  *changed = (before.cx != g->carry.cx || before.cy != g->carry.cy || before.prob != g->carry.prob) ? 1 : 0;
  return SLAMHIP_OK;
}

int slamhip_gmapping_set_shard_chain(slamhip_gmapping *g, int on) {
  if (!g) return bad("null filter");
  if (g->pending) return bad("a step is in flight");
  g->shard_chain = on != 0;
  if (on) g->step_carry = g->carry;
  return SLAMHIP_OK;
}

int slamhip_gmapping_carry_commit(slamhip_gmapping *g, const slamhip_carry_record *all, int world) {
  if (!g || !all) return bad("null argument");
  // what the next step starts from: the final entry of the last shard that matched
  for (int r = 0; r < world; ++r)
    if (all[r].has_active) g->step_carry = GmCarry{all[r].carry_cx, all[r].carry_cy, all[r].carry_prob};
  return SLAMHIP_OK;
}

int slamhip_gmapping_plan_resample(slamhip_gmapping *g, const double *all_raw_weights,
                                   uint32_t resample_seed, int *required, unsigned *idx_out) {
  if (!g || !all_raw_weights || !required) return bad("null argument");
  // normalize_weights over ALL particles in particle order: every rank does the same sums
  double total = 0;
  for (int i = 0; i < g->n_total; ++i) total += all_raw_weights[i];
  for (int i = 0; i < g->n_total; ++i) g->all_w[i] = all_raw_weights[i] / total;
  for (int i = 0; i < g->count; ++i) g->p[i].weight = g->all_w[g->first + i];
  *required = 0;
  if (g->traversed[0] * g->traversed[0] + g->traversed[1] * g->traversed[1] <= 0.5 &&
      std::fabs(double(g->traversed[2] <= 0.2)))
    return SLAMHIP_OK;
  int req = 0;
  int rc = slamhip_pf_resampling_is_required(g->n_total, g->all_w.data(), &req);
  if (rc || !req) return rc;
  if (!idx_out) return bad("resampling is required: idx_out is null");
  rc = slamhip_pf_resample(g->n_total, g->all_w.data(), resample_seed, idx_out);
  if (rc) return rc;
  *required = 1;
  return SLAMHIP_OK;
}

int slamhip_gmapping_export(slamhip_gmapping *g, void *blobs_out) {
  if (!g || !blobs_out) return bad("null argument");
  std::memcpy(blobs_out, g->p.data(), sizeof(GmParticle) * g->count);
  return SLAMHIP_OK;
}

int slamhip_gmapping_import(slamhip_gmapping *g, const void *all_blobs, const unsigned *idx) {
  if (!g || !all_blobs || !idx) return bad("null argument");
  const GmParticle *all = static_cast<const GmParticle *>(all_blobs);
  const int n = g->n_total;
  std::vector<char> seen(n, 0), dup(n, 0);
  for (int i = 0; i < n; ++i) {
    if (idx[i] >= (unsigned)n) return bad("resampling index out of range");
    dup[i] = seen[idx[i]];
    seen[idx[i]] = 1;
  }
  // weights of the new set, renormalised in particle order (particle_filter.h:104)
  std::vector<double> w(n);
  double total = 0;
  for (int i = 0; i < n; ++i) {
    w[i] = all[idx[i]].weight;
    total += w[i];
  }
  for (int i = 0; i < n; ++i) w[i] = w[i] / total;
  bool has_master = false;
  for (int i = 0; i < n; ++i) has_master |= (!dup[i] && all[idx[i]].is_master);
  const int hv = heaviest(w);
  std::vector<GmParticle> np(g->count);
  for (int l = 0; l < g->count; ++l) {
    const int i = g->first + l;
    if (dup[i])
      copy_as_duplicate(np[l], all[idx[i]]);
    else
      np[l] = all[idx[i]];
    np[l].weight = w[i];
    if (!has_master && i == hv) mark_master(np[l]);  // ensure_master_exists
  }
  g->p.swap(np);
  g->all_w = w;
  g->traversed[0] = g->traversed[1] = g->traversed[2] = 0;
  if (g->tp && !g->maps_handled_by_caller) {
    // `*new_particle = *sampled` copies the map too: with the tiled map that is a table copy, the tiles
    // get shared until one of the copies writes (lazy_tiled_grid_map.h:40-45,57-71)
    if (g->count != g->n_total)
      return bad("a sharded filter with per-particle maps resamples through slamhip_gmapping_import_maps");
    std::vector<int> src(n);
    for (int i = 0; i < n; ++i) src[i] = (int)idx[i];
    int rc = tile_pool_assign(g->tp, src.data());
    if (rc) return rc;
  }
  return SLAMHIP_OK;
}

int slamhip_gmapping_particle_map_export_size(slamhip_gmapping *g, int particle, size_t *bytes) {
  if (!g || !g->tp || !bytes) return bad("per-particle maps are not enabled");
  if (particle < 0 || particle >= g->count) return bad("particle index out of range");
  *bytes = tile_pool_export_size(g->tp, particle);
  return SLAMHIP_OK;
}

int slamhip_gmapping_particle_map_export(slamhip_gmapping *g, int particle, void *host_buf, size_t cap) {
  if (!g || !g->tp) return bad("per-particle maps are not enabled");
  if (particle < 0 || particle >= g->count) return bad("particle index out of range");
  SLAMHIP_CHECK(hipSetDevice(g->ctx->device));
  return tile_pool_export(g->tp, particle, host_buf, cap);
}

// remote_bodies == null: remote_bufs[k] is header + body in one host buffer (the C-ABI form); otherwise the headers
// (host) and the bodies (host or device memory) apart
static int import_maps_impl(slamhip_gmapping *g, const void *all_blobs, const unsigned *idx, int n_remote,
                            const int *remote_src, const void *const *remote_bufs, const void *const *remote_bodies) {
  // where every new local particle's map comes from: an old local slot, or one of the exported maps
  std::vector<int> src(g->count);
  for (int l = 0; l < g->count; ++l) {
    const int j = (int)idx[g->first + l];
    if (j >= g->first && j < g->first + g->count) {
      src[l] = j - g->first;
      continue;
    }
    int k = 0;
    while (k < n_remote && remote_src[k] != j) ++k;
    if (k == n_remote) return bad("the map of a particle resampled from another rank was not supplied");
    src[l] = -k - 1;
  }
  g->maps_handled_by_caller = true;
  int rc = slamhip_gmapping_import(g, all_blobs, idx);
  g->maps_handled_by_caller = false;
  if (rc) return rc;
  return remote_bodies ? tile_pool_assign_mixed_split(g->tp, src.data(), n_remote, remote_bufs, remote_bodies)
                       : tile_pool_assign_mixed(g->tp, src.data(), n_remote, remote_bufs);
}

int slamhip_gmapping_import_maps(slamhip_gmapping *g, const void *all_blobs, const unsigned *idx, int n_remote,
                                 const int *remote_src, const void *const *remote_bufs) {
  if (!g || !g->tp) return bad("per-particle maps are not enabled");
  if (!all_blobs || !idx || n_remote < 0 || (n_remote > 0 && (!remote_src || !remote_bufs))) return bad("null argument");
  SLAMHIP_CHECK(hipSetDevice(g->ctx->device));
  return import_maps_impl(g, all_blobs, idx, n_remote, remote_src, remote_bufs, nullptr);
}

int slamhip_gmapping_enable_particle_maps(slamhip_gmapping *g, int map_id, const slamhip_scan_adder_cfg *cfg,
                                          int extent_tiles, int pool_tiles) {
  if (!g || !cfg) return bad("null argument");
  if (!g->ctx) {
    set_error("this filter was created without a GPU context");
    return SLAMHIP_ERR_NO_DEVICE;
  }
  if (map_id < 0 || map_id >= (int)g->ctx->maps.size() || !g->ctx->maps[map_id].bound) return bad("unknown map id");
  if (extent_tiles <= 0 || pool_tiles < 2) return bad("bad tile pool shape");
  SLAMHIP_CHECK(hipSetDevice(g->ctx->device));
  const DeviceMap &m = g->ctx->maps[map_id];
  if (g->tp) tile_pool_destroy(g->tp);
  g->tp = nullptr;
  // one slot per LOCAL particle: a shard of the filter holds the maps of its own particles only
  int rc = tile_pool_create(g->ctx, g->count, extent_tiles, extent_tiles, m.scale, m.unknown, pool_tiles, &g->tp);
  if (rc) return rc;
  rc = tile_pool_init_from_dense(g->tp, m);
  if (rc) {
    tile_pool_destroy(g->tp);
    g->tp = nullptr;
    return rc;
  }
  g->upd = *cfg;
  g->update = false;  // the sequential shared-map mode and this one exclude each other
  bind_tiled_target(g);
  return SLAMHIP_OK;
}

int slamhip_gmapping_particle_map_download(slamhip_gmapping *g, int particle, int x0, int y0, int w, int h,
                                           double *payload3, double *aux2) {
  if (!g || !g->tp) return bad("per-particle maps are not enabled");
  if (particle < 0 || particle >= g->count) return bad("particle index out of range");
  return tile_pool_download(g->tp, particle, x0, y0, w, h, payload3, aux2);
}

int slamhip_gmapping_particle_maps_append(slamhip_gmapping *g, int n_jobs, const int *particles,
                                          const double *poses3, int n_raw, const double *range,
                                          const double *angle, const int *is_occ, long long *n_updates) {
  if (!g || !g->tp) return bad("per-particle maps are not enabled");
  if (n_updates) *n_updates = 0;
  if (n_jobs <= 0 || n_raw <= 0) return SLAMHIP_OK;
  if (!particles || !poses3 || !range || !angle) return bad("null argument");
  for (int k = 0; k < n_jobs; ++k) {
    if (particles[k] < 0 || particles[k] >= g->count) return bad("particle index out of range");
    for (int q = 0; q < k; ++q)
      if (particles[q] == particles[k]) return bad("a particle's map takes one scan per batch");
  }
  SLAMHIP_CHECK(hipSetDevice(g->ctx->device));
  raw_trig(g, n_raw, angle);
  slamhip_scan_adder_cfg cfg = g->upd;
  cfg.rule = SLAMHIP_RULE_GMAPPING;
  cfg.scan_quality = 1.0;
  long long nu = 0;
  int rc = mu_append_batch(g->ctx, g->tp, &cfg, n_jobs, poses3, particles, n_raw, range, g->trig_cos.data(),
                           g->trig_sin.data(), is_occ, &nu);
  if (rc) return rc;
  g->cell_updates += nu;
  if (n_updates) *n_updates = nu;
  return SLAMHIP_OK;
}

int slamhip_gmapping_particle_map_stats(slamhip_gmapping *g, long long *tiles_in_use, long long *tiles_shared,
                                        long long *bytes, long long *cow_copies, long long *cell_updates) {
  if (!g || !g->tp) return bad("per-particle maps are not enabled");
  tile_pool_stats(g->tp, tiles_in_use, tiles_shared, bytes, cow_copies);
  if (cell_updates) *cell_updates = g->cell_updates;
  return SLAMHIP_OK;
}

int slamhip_gmapping_step(slamhip_gmapping *g, int map_id, int n_raw, const double *range,
                          const double *angle, const int *is_occ, const double odom_delta[3],
                          uint32_t resample_seed, int *resampled, unsigned *idx_out) {
  if (!g) return bad("null filter");
  if (g->count != g->n_total) return bad("slamhip_gmapping_step needs the whole filter on one context");
  std::vector<double> raw(g->n_total);
  int rc = slamhip_gmapping_predict_match(g, map_id, n_raw, range, angle, is_occ, odom_delta, raw.data());
  if (rc) return rc;
  std::vector<unsigned> idx(g->n_total);
  int req = 0;
  rc = slamhip_gmapping_plan_resample(g, raw.data(), resample_seed, &req, idx.data());
  if (rc) return rc;
  if (req) {
    std::vector<GmParticle> all(g->p);
    rc = slamhip_gmapping_import(g, all.data(), idx.data());
    if (rc) return rc;
    if (idx_out) std::memcpy(idx_out, idx.data(), sizeof(unsigned) * g->n_total);
  }
  if (resampled) *resampled = req;
  return SLAMHIP_OK;
}

// what match_finish will hand out as raw weights, without touching the particles (gmapping_world.h:99: weight *= the
// match's probability for the particles that matched)
static void provisional_weights(const slamhip_gmapping *g, double *out) {
  for (int i = 0; i < g->count; ++i) out[i] = g->p[i].weight;
  for (size_t k = 0; k < g->act_idx.size(); ++k) out[g->act_idx[k]] = g->act[k]->best_prob * g->p[g->act_idx[k]].weight;
}

// true when some shard's first job met the cache entry handed to it with another value than it ran with (none):
// slamhip_gmapping_carry_fix's condition for a first round, evaluated for every shard from the records alone
static bool carry_repair_needed(const GmCarry &step_carry, const slamhip_carry_record *all, int world) {
  GmCarry pred = step_carry;
  for (int r = 0; r < world; ++r) {
    if (!all[r].has_active) continue;
    if (pred.prob != -1.0 && all[r].first_cx == pred.cx && all[r].first_cy == pred.cy && pred.prob != all[r].first_v0)
      return true;
    pred = GmCarry{all[r].carry_cx, all[r].carry_cy, all[r].carry_prob};
  }
  return false;
}

// leaves a step that cannot go on: the filter is usable again (the particles keep whatever the step had already
// done to them -- odometry, pose noise -- but no match result is applied)
static void match_abort(slamhip_gmapping *g) {
  g->pending = false;
  g->act.clear();
  g->act_idx.clear();
}

int slamhip_gmapping_match_abort(slamhip_gmapping *g) {
  if (!g) return bad("null filter");
  match_abort(g);
  return SLAMHIP_OK;
}

// Resampling of a sharded filter with per-particle maps, inside the library: `*new_particle = *sampled`
// (particle_filter.h:92-96) for a particle that lives on another rank means its map travels -- tile by tile, like the
// copy-on-write copy of lazy_tiled_grid_map.h:40-71, except that the tiles cross xGMI.  Every rank reads the same plan
// off the resampling indices: which maps leave which rank for which.  Two small all-gathers (sizes, then the maps'
// headers: tile positions and ancestor ordinals), ONE point-to-point exchange of the tile contents device to device
// (slamhip_shard_exchange: RCCL send / recv in one group), then the import.
// *everyone_left: the migration failed on some rank and EVERY rank returned an error from the same collective
// (nothing to defer); otherwise an error is this rank's alone.
static int migrate_and_import(slamhip_gmapping *g, int rank, int world, const std::vector<GmParticle> &blobs,
                              const std::vector<unsigned> &idx, bool *everyone_left) {
  *everyone_left = false;
  slamhip_ctx *ctx = g->ctx;
  std::vector<int> start(world + 1, 0);
  for (int r = 0; r < world; ++r) start[r + 1] = start[r] + g->shard_counts[r];
  auto owner = [&](int j) {
    int r = 0;
    while (j >= start[r + 1]) ++r;
    return r;
  };
  // need[r]: sources rank r draws from other ranks (ascending, each once); exports[q]: sources rank q sends
  std::vector<std::vector<int>> need(world), exports(world);
  for (int r = 0; r < world; ++r) {
    std::vector<char> seen(g->n_total, 0);
    for (int j = start[r]; j < start[r + 1]; ++j) {
      const int src = (int)idx[j];
      if (owner(src) != r) seen[src] = 1;
    }
    for (int j = 0; j < g->n_total; ++j)
      if (seen[j]) need[r].push_back(j);
  }
  {
    std::vector<char> seen(g->n_total, 0);
    for (int r = 0; r < world; ++r)
      for (int j : need[r]) seen[j] = 1;
    for (int j = 0; j < g->n_total; ++j)
      if (seen[j]) exports[owner(j)].push_back(j);
  }
  size_t n_exports_all = 0;
  for (int q = 0; q < world; ++q) n_exports_all += exports[q].size();
  if (n_exports_all == 0)  // every new particle's source is local everywhere
    return import_maps_impl(g, blobs.data(), idx.data(), 0, nullptr, nullptr, nullptr);
  // A rank-local failure between the migration's collectives must not leave the others waiting in the next one:
  // it travels as a STATUS WORD in front of this rank's block of the next all-gather, and every rank leaves the
  // migration behind that collective, before anybody enters the exchange.
  // sizes of my exports, then everybody's
  const std::vector<int> &mine = exports[rank];
  std::vector<unsigned long long> my_sizes(1 + 2 * mine.size(), 0ull);  // [0]: status
  size_t my_header_bytes = 0, my_body_bytes = 0;
  for (size_t k = 0; k < mine.size(); ++k) {
    size_t hb = 0, bb = 0;
    tile_pool_export_sizes(g->tp, mine[k] - g->first, &hb, &bb);
    my_sizes[1 + 2 * k] = hb;
    my_sizes[2 + 2 * k] = bb;
    my_header_bytes += hb;
    my_body_bytes += bb;
  }
  int local_rc = SLAMHIP_OK;
  std::string local_msg;
  auto fail_locally = [&](int rc_) {
    if (!local_rc && rc_) {
      local_rc = rc_;
      local_msg = slamhip_last_error();
    }
  };
  // the send buffer grows BEFORE the first collective
  if (my_body_bytes > g->mig_send_cap) {
    hipError_t he = hipStreamSynchronize(ctx->stream);
    if (he == hipSuccess) {
      if (g->d_mig_send) hipFree(g->d_mig_send);
      g->d_mig_send = nullptr;
      g->mig_send_cap = 0;
      he = hipMalloc(&g->d_mig_send, my_body_bytes);
    }
    if (he != hipSuccess) fail_locally(hip_fail(he, "staging of the migrating maps (send)"));
    else g->mig_send_cap = my_body_bytes;
  }
  my_sizes[0] = (unsigned long long)(unsigned)local_rc;
  std::vector<int> cnt(world);
  for (int q = 0; q < world; ++q) cnt[q] = 1 + 2 * (int)exports[q].size();
  std::vector<unsigned long long> all_sizes((size_t)world + 2 * n_exports_all);
  int rc = slamhip_shard_allgather(ctx, my_sizes.data(), cnt.data(), (int)sizeof(unsigned long long), all_sizes.data());
  if (rc) return rc;
  auto leave_together = [&](int failed_rank, const char *where) {
    *everyone_left = true;
    if (local_rc) {
      set_error(local_msg);
      return local_rc;
    }
    set_error("rank " + std::to_string(failed_rank) + " of the shard group failed " + where +
              ": the resampling was abandoned on every rank");
    return (int)SLAMHIP_ERR_STATE;
  };
  // (rank, export ordinal) -> sizes and the offset of its header in the gathered header blob
  std::vector<std::vector<size_t>> hb_of(world), bb_of(world), hoff_of(world);
  size_t hoff = 0, at = 0;
  std::vector<int> hcnt(world, 0);
  std::vector<size_t> hbase(world, 0);
  int failed_rank = -1;
  for (int q = 0; q < world; ++q) {
    if (all_sizes[at] != 0ull && failed_rank < 0) failed_rank = q;
    ++at;
    hbase[q] = hoff;
    hoff += 8;  // the status word in front of rank q's headers
    hcnt[q] = 8;
    for (size_t k = 0; k < exports[q].size(); ++k, at += 2) {
      hb_of[q].push_back((size_t)all_sizes[at]);
      bb_of[q].push_back((size_t)all_sizes[at + 1]);
      hoff_of[q].push_back(hoff);
      hoff += hb_of[q].back();
      hcnt[q] += (int)hb_of[q].back();
    }
  }
  if (failed_rank >= 0) return leave_together(failed_rank, "while staging its migrating maps");
  // my maps: headers to host memory, tile contents into the device send buffer (queued on the stream)
  std::vector<char> my_headers(8 + my_header_bytes, 0);
  std::vector<size_t> send_off(mine.size());
  {
    size_t ho = 8, bo = 0;
    for (size_t k = 0; k < mine.size() && !local_rc; ++k) {
      send_off[k] = bo;
      if (debug_fail_now(g, 2)) fail_locally(SLAMHIP_ERR_STATE);
      else fail_locally(tile_pool_export_split(g->tp, mine[k] - g->first, my_headers.data() + ho, g->d_mig_send + bo, false));
      ho += (size_t)my_sizes[1 + 2 * k];
      bo += (size_t)my_sizes[2 + 2 * k];
    }
  }
  // the exchange: my sends by (destination, source) ascending, my receives by source ascending -- one order per pair
  auto ordinal = [&](int q, int src) {
    const std::vector<int> &e = exports[q];
    return (size_t)(std::lower_bound(e.begin(), e.end(), src) - e.begin());
  };
  // ... and the receive buffer grows before the second collective
  size_t recv_bytes = 0;
  std::vector<size_t> recv_off(need[rank].size());
  for (size_t k = 0; k < need[rank].size(); ++k) {
    const int src = need[rank][k], q = owner(src);
    recv_off[k] = recv_bytes;
    recv_bytes += bb_of[q][ordinal(q, src)];
  }
  if (recv_bytes > g->mig_recv_cap) {
    hipError_t he = hipStreamSynchronize(ctx->stream);
    if (he == hipSuccess) {
      if (g->d_mig_recv) hipFree(g->d_mig_recv);
      g->d_mig_recv = nullptr;
      g->mig_recv_cap = 0;
      he = hipMalloc(&g->d_mig_recv, recv_bytes);
    }
    if (he != hipSuccess) fail_locally(hip_fail(he, "staging of the migrating maps (receive)"));
    else g->mig_recv_cap = recv_bytes;
  }
  {
    const unsigned long long status = (unsigned long long)(unsigned)local_rc;
    std::memcpy(my_headers.data(), &status, 8);
  }
  std::vector<char> all_headers(hoff ? hoff : 1);
  rc = slamhip_shard_allgather(ctx, my_headers.data(), hcnt.data(), 1, all_headers.data());
  if (rc) return rc;
  for (int q = 0; q < world; ++q) {
    unsigned long long status = 0;
    std::memcpy(&status, all_headers.data() + hbase[q], 8);
    if (status != 0ull && failed_rank < 0) failed_rank = q;
  }
  if (failed_rank >= 0) return leave_together(failed_rank, "while exporting its migrating maps");
  std::vector<slamhip_shard_msg> sends, recvs;
  for (int r = 0; r < world; ++r) {
    if (r == rank) continue;
    for (int src : need[r]) {
      if (owner(src) != rank) continue;
      const size_t k = ordinal(rank, src);
      sends.push_back(slamhip_shard_msg{r, g->d_mig_send + send_off[k], (size_t)my_sizes[2 + 2 * k]});
      g->map_bytes_sent += (long long)my_sizes[2 + 2 * k];
    }
  }
  for (size_t k = 0; k < need[rank].size(); ++k) {
    const int src = need[rank][k], q = owner(src);
    recvs.push_back(slamhip_shard_msg{q, g->d_mig_recv + recv_off[k], bb_of[q][ordinal(q, src)]});
  }
  rc = slamhip_shard_exchange(ctx, (int)sends.size(), sends.data(), (int)recvs.size(), recvs.data());
  if (rc) return rc;
  g->maps_migrated += (long long)need[rank].size();
  std::vector<const void *> hdr(need[rank].size()), body(need[rank].size());
  for (size_t k = 0; k < need[rank].size(); ++k) {
    const int src = need[rank][k], q = owner(src);
    hdr[k] = all_headers.data() + hoff_of[q][ordinal(q, src)];
    body[k] = g->d_mig_recv + recv_off[k];
  }
  // (behind the migration's last collective: a failure here is this rank's alone -- the caller defers it to the next
  // step's status word)
  if (debug_fail_now(g, 3)) return SLAMHIP_ERR_STATE;
  return import_maps_impl(g, blobs.data(), idx.data(), (int)need[rank].size(), need[rank].data(), hdr.data(), body.data());
}

int slamhip_gmapping_step_sharded(slamhip_gmapping *g, int map_id, int n_raw, const double *range,
                                  const double *angle, const int *is_occ, const double odom_delta[3],
                                  uint32_t resample_seed, int *resampled, unsigned *idx_out) {
  if (!g) return bad("null filter");
  if (!g->ctx) return bad("a sharded step needs a GPU context");
  if (g->update) return bad("the shared-map update is sequential over all particles and cannot be sharded");
  int rank = 0, world = 1;
  int rc = slamhip_shard_info(g->ctx, &rank, &world);
  if (rc) return rc;
  // contiguous blocks in rank order: every rank learns the others' block sizes once
  if ((int)g->shard_counts.size() != world) {
    std::vector<int> ones(world, 1), cnt(world, 0);
    const int mine = g->count;
    rc = slamhip_shard_allgather(g->ctx, &mine, ones.data(), (int)sizeof(int), cnt.data());
    if (rc) return rc;
    int at = 0;
    for (int r = 0; r < world; ++r) {
      if (r == rank && at != g->first) return bad("shards are not contiguous blocks in rank order");
      at += cnt[r];
    }
    if (at != g->n_total) return bad("the shards do not add up to n_total particles");
    g->shard_counts = cnt;
    rc = slamhip_gmapping_set_shard_chain(g, 1);
    if (rc) return rc;
  }
  // A failure of this rank must not leave the others waiting in a collective: whatever happens locally up to the
  // step's first all-gather -- and whatever happened behind the last collective of the step before -- travels in a
  // status word next to the carry record, and EVERY rank abandons the step when any word is set.
  int local_rc = g->deferred_rc;
  g->deferred_rc = 0;
  if (!local_rc) local_rc = slamhip_gmapping_match_begin(g, map_id, n_raw, range, angle, is_occ, odom_delta);
  const std::string local_msg = local_rc ? std::string(slamhip_last_error()) : std::string();
  // ONE collective in the common case: every shard sends its carry record together with the raw weights its
  // particles will have if no cache hand-over needs repair.  Whether one does is a pure function of the records --
  // shard r re-matches its first job only when the final entry of the nearest matching shard before it (or of
  // the previous step) hits that job's first run with another value -- so every rank reaches the same verdict
  // without exchanging flags; only then the slow protocol (exchange, re-match, until no final entry changes)
  // and a second all-gather of the weights run.
  std::vector<slamhip_carry_record> recs(world);
  std::vector<double> raw(g->count), all(g->n_total);
  const std::vector<int> ones(world, 1);
  bool settled = false;
  int late_rc = 0;  // settled path: this rank failed behind the step's first collective
  std::string late_msg;
  {
    slamhip_carry_record mine;
    std::memset(&mine, 0, sizeof(mine));
    if (!local_rc) {
      local_rc = slamhip_gmapping_carry_record(g, &mine);
      if (local_rc) std::memset(&mine, 0, sizeof(mine));
    }
    constexpr int kRecDoubles = 1 + (int)((sizeof(slamhip_carry_record) + 7) / 8);  // status word first
    std::vector<int> cnt(world);
    int total = 0;
    for (int r = 0; r < world; ++r) {
      cnt[r] = kRecDoubles + g->shard_counts[r];
      total += cnt[r];
    }
    std::vector<double> blk(cnt[rank], 0.0), gathered(total, 0.0);
    blk[0] = (double)local_rc;
    std::memcpy(blk.data() + 1, &mine, sizeof(mine));
    if (!local_rc) provisional_weights(g, blk.data() + kRecDoubles);
    rc = slamhip_shard_allgather(g->ctx, blk.data(), cnt.data(), (int)sizeof(double), gathered.data());
    if (rc) {
      match_abort(g);
      return rc;
    }
    int at = 0, wat = 0, failed_rank = -1;
    for (int r = 0; r < world; ++r) {
      if (gathered[at] != 0.0 && failed_rank < 0) failed_rank = r;
      std::memcpy(&recs[r], gathered.data() + at + 1, sizeof(slamhip_carry_record));
      std::memcpy(all.data() + wat, gathered.data() + at + kRecDoubles, sizeof(double) * g->shard_counts[r]);
      at += cnt[r];
      wat += g->shard_counts[r];
    }
    if (failed_rank >= 0) {
      match_abort(g);
      if (local_rc) {
        set_error(local_msg);
        return local_rc;
      }
      set_error("rank " + std::to_string(failed_rank) + " of the shard group failed: the step was abandoned on every rank");
      return SLAMHIP_ERR_STATE;
    }
    if (!carry_repair_needed(g->step_carry, recs.data(), world)) {
      int changed = 0;
      rc = slamhip_gmapping_carry_fix(g, recs.data(), world, rank, &changed);  // (passes the cache through; no re-match)
      if (!rc && changed) rc = bad("internal: a shard re-matched although no cache hand-over needed repair");
      // a failure from here on is behind the step's first collective.  The others cannot see it yet, and they WILL
      // enter the resampling's collectives when the weights say so: this rank goes there with them (late_rc)
      if (rc) {
        late_rc = rc;
        late_msg = slamhip_last_error();
      }
      settled = true;
    }
  }
  if (!settled) {
    // the shared OOPE cache across shards: exchange, re-check, until no shard changes its final entry.  The flag a
    // rank contributes is 1 when its entry changed, -1 when it failed: then every rank stops.
    for (int round = 0; round <= world; ++round) {
      int flag = 0;
      if (round > 0) {
        slamhip_carry_record mine;
        std::memset(&mine, 0, sizeof(mine));
        rc = slamhip_gmapping_carry_record(g, &mine);
        if (rc) flag = -1;
        rc = slamhip_shard_allgather(g->ctx, &mine, ones.data(), (int)sizeof(mine), recs.data());
        if (rc) {
          match_abort(g);
          return rc;
        }
      }
      int changed = 0;
      if (flag == 0) {
        local_rc = slamhip_gmapping_carry_fix(g, recs.data(), world, rank, &changed);
        flag = local_rc ? -1 : changed;
      }
      const std::string msg = flag < 0 ? std::string(slamhip_last_error()) : std::string();
      int any = 0, failed_rank = -1;
      std::vector<int> flags(world, 0);
      rc = slamhip_shard_allgather(g->ctx, &flag, ones.data(), (int)sizeof(int), flags.data());
      if (rc) {
        match_abort(g);
        return rc;
      }
      for (int r = 0; r < world; ++r) {
        if (flags[r] < 0 && failed_rank < 0) failed_rank = r;
        any |= flags[r];
      }
      if (failed_rank >= 0) {
        match_abort(g);
        if (flag < 0) {
          set_error(msg);
          return local_rc ? local_rc : SLAMHIP_ERR_STATE;
        }
        set_error("rank " + std::to_string(failed_rank) + " of the shard group failed while repairing the cache hand-over");
        return SLAMHIP_ERR_STATE;
      }
      if (!any) break;
    }
  }
  rc = late_rc;
  if (!rc) rc = slamhip_gmapping_carry_commit(g, recs.data(), world);
  if (!rc) rc = slamhip_gmapping_match_finish(g, raw.data());
  if (rc && settled && !late_rc) {
    late_rc = rc;
    late_msg = slamhip_last_error();
  }
  if (late_rc) match_abort(g);
  if (!settled) {
    // a re-match may have changed a shard's weights: all raw weights once more, in particle order (with a status
    // word in front: match_finish may have failed on some rank)
    std::vector<int> cnt(world);
    int total = 0;
    for (int r = 0; r < world; ++r) {
      cnt[r] = 1 + g->shard_counts[r];
      total += cnt[r];
    }
    std::vector<double> blk(cnt[rank], 0.0), gathered(total, 0.0);
    blk[0] = (double)rc;
    const std::string msg = rc ? std::string(slamhip_last_error()) : std::string();
    std::memcpy(blk.data() + 1, raw.data(), sizeof(double) * g->count);
    int rc2 = slamhip_shard_allgather(g->ctx, blk.data(), cnt.data(), (int)sizeof(double), gathered.data());
    if (rc2) {
      match_abort(g);
      return rc2;
    }
    int at = 0, wat = 0, failed_rank = -1;
    for (int r = 0; r < world; ++r) {
      if (gathered[at] != 0.0 && failed_rank < 0) failed_rank = r;
      std::memcpy(all.data() + wat, gathered.data() + at + 1, sizeof(double) * g->shard_counts[r]);
      at += cnt[r];
      wat += g->shard_counts[r];
    }
    if (failed_rank >= 0) {
      match_abort(g);
      if (rc) {
        set_error(msg);
        return rc;
      }
      set_error("rank " + std::to_string(failed_rank) + " of the shard group failed in its map update");
      return SLAMHIP_ERR_STATE;
    }
  }
  std::vector<unsigned> idx(g->n_total);
  int req = 0;
  // (same inputs, same verdict everywhere -- also on a rank that failed late: the plan needs the gathered weights only)
  rc = slamhip_gmapping_plan_resample(g, all.data(), resample_seed, &req, idx.data());
  if (rc) {
    g->deferred_rc = rc;
    return rc;
  }
  if (!req) {
    if (late_rc) {  // no collective follows in this step: the others learn of it at the next one's
      g->deferred_rc = late_rc;
      set_error(late_msg);
      return late_rc;
    }
    if (resampled) *resampled = 0;
    return SLAMHIP_OK;
  }
  {
    // the particle records of all ranks, each rank's block headed by a status word: a rank that failed late (its map
    // update, say) is in this collective with the others, and EVERY rank abandons the resampling together
    std::vector<GmParticle> blobs(g->n_total);
    std::vector<int> bcnt(world);
    size_t total = 0;
    for (int r = 0; r < world; ++r) {
      bcnt[r] = 8 + (int)sizeof(GmParticle) * g->shard_counts[r];
      total += (size_t)bcnt[r];
    }
    std::vector<char> blk((size_t)bcnt[rank], 0), gathered(total, 0);
    const unsigned long long status = (unsigned long long)(unsigned)late_rc;
    std::memcpy(blk.data(), &status, 8);
    std::memcpy(blk.data() + 8, g->p.data(), sizeof(GmParticle) * g->count);
    rc = slamhip_shard_allgather(g->ctx, blk.data(), bcnt.data(), 1, gathered.data());
    if (rc) return rc;
    size_t at = 0;
    int pat = 0, failed_rank = -1;
    for (int r = 0; r < world; ++r) {
      unsigned long long st_r = 0;
      std::memcpy(&st_r, gathered.data() + at, 8);
      if (st_r != 0ull && failed_rank < 0) failed_rank = r;
      std::memcpy(blobs.data() + pat, gathered.data() + at + 8, sizeof(GmParticle) * g->shard_counts[r]);
      at += (size_t)bcnt[r];
      pat += g->shard_counts[r];
    }
    if (failed_rank >= 0) {
      if (late_rc) {
        set_error(late_msg);
        return late_rc;
      }
      set_error("rank " + std::to_string(failed_rank) + " of the shard group failed behind the step's first collective: "
                "the resampling was abandoned on every rank");
      return SLAMHIP_ERR_STATE;
    }
    // own maps: the maps of particles drawn from other ranks travel first (two all-gathers and one exchange that
    // every rank enters, whatever it needs itself; collective-complete on failure, see migrate_and_import)
    bool everyone_left = false;
    rc = g->tp ? migrate_and_import(g, rank, world, blobs, idx, &everyone_left)
               : slamhip_gmapping_import(g, blobs.data(), idx.data());
    if (rc) {
      if (!everyone_left) g->deferred_rc = rc;  // (this rank's alone: the others learn of it at the next step's collective)
      return rc;
    }
    if (idx_out) std::memcpy(idx_out, idx.data(), sizeof(unsigned) * g->n_total);
  }
  if (resampled) *resampled = req;
  return SLAMHIP_OK;
}

#ifdef SLAMHIP_TESTING
// testing aid, not part of include/slamhip.h (see slamhip_gmapping::debug_fail_where)
int slamhip_gmapping_debug_fail(slamhip_gmapping *g, int where, int nth_call) {
  if (!g) return bad("null filter");
  g->debug_fail_where = where;
  g->debug_fail_countdown = nth_call;
  return SLAMHIP_OK;
}
// testing aid: the in-tile neighbourhood masks of the filter's per-particle maps (tile_pool.h) -- *valid: whether the
// pool holds masks; *mismatches: cells whose stored mask differs from the one the cells of their tile give
int slamhip_gmapping_debug_nbr_masks(slamhip_gmapping *g, int *valid, long long *mismatches) {
  if (!g || !g->tp) return bad("no per-particle maps");
  if (valid) *valid = g->tp->nbr_ok ? 1 : 0;
  long long n = 0;
  const int rc = tile_pool_nbr_check(g->tp, &n);
  if (mismatches) *mismatches = n;
  return rc;
}
// ... and of the pool's settle states (tile_pool.h TilePool::d_state): words that differ from what the payloads give
int slamhip_gmapping_debug_settle_states(slamhip_gmapping *g, long long *mismatches) {
  if (!g || !g->tp || !mismatches) return bad("no per-particle maps");
  return tile_pool_state_check(g->tp, mismatches);
}
#endif  // SLAMHIP_TESTING

int slamhip_gmapping_migration_stats(slamhip_gmapping *g, long long *maps_received, long long *tile_bytes_sent) {
  if (!g) return bad("null filter");
  if (maps_received) *maps_received = g->maps_migrated;
  if (tile_bytes_sent) *tile_bytes_sent = g->map_bytes_sent;
  return SLAMHIP_OK;
}

int slamhip_gmapping_set_map_update(slamhip_gmapping *g, const slamhip_scan_adder_cfg *cfg) {
  if (!g) return bad("null filter");
  if (cfg && g->tp) return bad("per-particle maps are enabled: the shared-map update mode excludes them");
  if (cfg && g->count != g->n_total)
    return bad("the map update inside the step needs the whole filter on one context: the reference's "
               "particles share one map and update it one after another");
  g->update = cfg != nullptr;
  if (cfg) g->upd = *cfg;
  return SLAMHIP_OK;
}

int slamhip_gmapping_set(slamhip_gmapping *g, const double *poses, const double *weights) {
  if (!g) return bad("null filter");
  for (int i = 0; i < g->count; ++i) {
    if (poses) std::memcpy(g->p[i].pose, poses + 3 * i, sizeof(double) * 3);
    if (weights) g->p[i].weight = weights[i];
  }
  return SLAMHIP_OK;
}

int slamhip_gmapping_get(slamhip_gmapping *g, double *poses, double *weights, int *is_master) {
  if (!g) return bad("null filter");
  for (int i = 0; i < g->count; ++i) {
    if (poses) std::memcpy(poses + 3 * i, g->p[i].pose, sizeof(double) * 3);
    if (weights) weights[i] = g->p[i].weight;
    if (is_master) is_master[i] = g->p[i].is_master;
  }
  return SLAMHIP_OK;
}

int slamhip_gmapping_stats(slamhip_gmapping *g, long long *scorer_calls, long long *poses_evaluated,
                           long long *launches, long long *carry_reruns) {
  if (!g) return bad("null filter");
  if (scorer_calls) *scorer_calls = g->scorer_calls;
  if (poses_evaluated) *poses_evaluated = g->poses_evaluated;
  if (launches) *launches = g->launches;
  if (carry_reruns) *carry_reruns = g->carry_reruns;
  return SLAMHIP_OK;
}

}  // extern "C"
