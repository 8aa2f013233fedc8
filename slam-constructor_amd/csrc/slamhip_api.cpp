// slamhip_api.cpp -- host side of the C-ABI (include/slamhip.h): context, map mirror, scan
// upload, batched scoring, host-resident pieces of the path (filter_scan, weights, particle
// filter bookkeeping).  Device work is in score_kernels.hip; matchers in matchers.cpp.
//
// There is deliberately no CPU implementation of the scorer here: if HIP is unavailable every
// entry point that needs the GPU returns SLAMHIP_ERR_NO_DEVICE / SLAMHIP_ERR_HIP.

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <random>

#include "slamhip_internal.h"
#include "libm_exact.h"

namespace slamhip {

static thread_local std::string g_last_error;

void set_error(const std::string &msg) { g_last_error = msg; }

int hip_fail(hipError_t e, const char *what) {
  g_last_error = std::string(what) + ": " + hipGetErrorString(e);
  return e == hipErrorNoDevice ? SLAMHIP_ERR_NO_DEVICE : SLAMHIP_ERR_HIP;
}

static int invalid(const char *msg) {
  g_last_error = msg;
  return SLAMHIP_ERR_INVALID;
}

// staging buffers are pinned, mapped into the GPU's address space and coherent: kernels read
// poses from them and write scores into them directly (no copy engine round trip)
static constexpr unsigned kPinned = hipHostMallocMapped | hipHostMallocCoherent;

static DeviceMap *get_map(slamhip_ctx *ctx, int map_id) {
  if (!ctx || map_id < 0 || map_id >= (int)ctx->maps.size() || !ctx->maps[map_id].bound)
    return nullptr;
  return &ctx->maps[map_id];
}

int ensure_pose_capacity(slamhip_ctx *ctx, int n) {
  if (n <= ctx->pose_cap) return SLAMHIP_OK;
  int cap = 256;
  while (cap < n) cap *= 2;
  // kernels of BOTH launch lanes read and write the staging buffers
  SLAMHIP_CHECK(hipStreamSynchronize(ctx->stream));
  if (ctx->stream_b) SLAMHIP_CHECK(hipStreamSynchronize(ctx->stream_b));
  if (ctx->d_poses) {
    hipFree(ctx->d_poses);
    hipFree(ctx->d_scores);
    hipFree(ctx->d_pose_sc);
    hipFree(ctx->d_gm_info);
    hipHostFree(ctx->h_poses);
    hipHostFree(ctx->h_scores);
    hipHostFree(ctx->h_pose_sc);
    hipHostFree(ctx->h_gm_info);
    hipHostFree(ctx->h_pose_slot);
    hipHostFree(ctx->h_fprints);
    ctx->h_fprints = nullptr;
    // a failed allocation below must not leave pointers that ctx_destroy (or the next call) frees again
    ctx->d_poses = ctx->d_scores = ctx->d_pose_sc = nullptr;
    ctx->d_gm_info = nullptr;
    ctx->h_poses = ctx->h_scores = ctx->h_pose_sc = nullptr;
    ctx->h_gm_info = nullptr;
    ctx->h_pose_slot = nullptr;
    ctx->pose_cap = 0;
  }
  SLAMHIP_CHECK(hipHostMalloc(&ctx->h_pose_slot, sizeof(int) * cap, kPinned));
  SLAMHIP_CHECK(hipHostMalloc(&ctx->h_fprints, sizeof(unsigned long long) * cap, kPinned));
  SLAMHIP_CHECK(hipMalloc(&ctx->d_poses, sizeof(double) * 3 * cap));
  SLAMHIP_CHECK(hipMalloc(&ctx->d_scores, sizeof(double) * cap));
  SLAMHIP_CHECK(hipMalloc(&ctx->d_pose_sc, sizeof(double) * 2 * cap));
  SLAMHIP_CHECK(hipMalloc(&ctx->d_gm_info, sizeof(GmPoseInfo) * cap));
  SLAMHIP_CHECK(hipHostMalloc(&ctx->h_poses, sizeof(double) * 3 * cap, kPinned));
  SLAMHIP_CHECK(hipHostMalloc(&ctx->h_scores, sizeof(double) * cap, kPinned));
  SLAMHIP_CHECK(hipHostMalloc(&ctx->h_pose_sc, sizeof(double) * 2 * cap, kPinned));
  SLAMHIP_CHECK(hipHostMalloc(&ctx->h_gm_info, sizeof(GmPoseInfo) * cap, kPinned));
  ctx->pose_cap = cap;
  return SLAMHIP_OK;
}

static int check_cfg(const DeviceMap &m, const slamhip_spe_cfg *cfg) {
  if (!cfg) return invalid("null spe cfg");
  if (cfg->oope == SLAMHIP_OOPE_GMAPPING) {
    if (m.cell_model != SLAMHIP_CELL_GMAPPING)
      return invalid("GMAPPING OOPE needs a SLAMHIP_CELL_GMAPPING map");
    if (cfg->gm_window < 0 || cfg->gm_window > 4) return invalid("gm_window out of range");
    // (SLAMHIP_SUM_SEQUENTIAL: K3 sums in the canonical tree order and patches cross-pose cache hits afterwards --
    // neither is the reference's beam-order sum; since r06 that order is served by k_score_gmapping_exact, the plain
    // restatement with glibc's exp and the cache on the device: score_exact below)
    return SLAMHIP_OK;
  }
  if (cfg->oope < SLAMHIP_OOPE_OBSTACLE || cfg->oope > SLAMHIP_OOPE_GMAPPING)
    return invalid("unknown OOPE kind");
  if (cfg->oope != SLAMHIP_OOPE_OBSTACLE) {
    if (!(cfg->area[0] <= cfg->area[1] && cfg->area[2] <= cfg->area[3]))
      return invalid("sp_analysis_area needs bot <= top and left <= right");
    const double cells = ((cfg->area[1] - cfg->area[0]) / m.scale + 2) * ((cfg->area[3] - cfg->area[2]) / m.scale + 2);
    if (!(cells < 1e4)) {
      g_last_error = "sp_analysis_area covers more than 10^4 cells per beam";
      return SLAMHIP_ERR_UNSUPPORTED;
    }
  }
  if (m.cell_model == SLAMHIP_CELL_GMAPPING)
    return invalid("obstacle OOPE over a GMAPPING payload: upload prob_occ as an OCC map");
  if (m.cell_model == SLAMHIP_CELL_TBM && cfg->oie != SLAMHIP_OIE_DISCREPANCY)
    return invalid("OccupancyOIE over TBM cells: upload occupancy().prob_occ as an OCC map");
  return SLAMHIP_OK;
}

// The neighbourhood masks of a dense GMAPPING window for threshold th (MapView::nbr_ok), derived on the context's stream
// by the first GMapping scorer call that finds none (or finds another threshold's); from then on the map's writers
// keep them.  The pass reads nine occupancies per cell (0.5 GB of cells: a fraction of a millisecond) and is waited
// for: the scorer's second launch lane and the chains' own streams are not ordered behind this one.
static int map_nbr_masks(slamhip_ctx *ctx, DeviceMap &m, double th) {
  if (m.nbr_ok && m.nbr_th == th) return SLAMHIP_OK;
  SLAMHIP_CHECK(launch_nbr_build(m.d_payload, m.width, m.height, m.pitch, th, 0, 0, m.width, m.height, ctx->stream));
  SLAMHIP_CHECK(hipStreamSynchronize(ctx->stream));
  m.nbr_th = th;
  m.nbr_ok = true;
  return SLAMHIP_OK;
}

// The probability plane of a dense TBM window (DeviceMap::d_prob): derived on the context's stream by the first
// 1-cell scorer call that finds none, and waited for (like the masks above); from then on the map's writers keep it.
static int map_prob_plane(slamhip_ctx *ctx, DeviceMap &m) {
  if (m.prob_ok) return SLAMHIP_OK;
  if (!m.d_prob) SLAMHIP_CHECK(hipMalloc(&m.d_prob, sizeof(double) * (size_t)m.pitch * m.height));
  SLAMHIP_CHECK(launch_prob_build(m.d_payload, m.d_prob, m.width, m.height, m.pitch, 0, 0, m.width, m.height, ctx->stream));
  SLAMHIP_CHECK(hipStreamSynchronize(ctx->stream));
  m.prob_ok = true;
  return SLAMHIP_OK;
}
// whether a scorer configuration over this map reads the plane: the obstacle OOPE under the discrepancy OIE (the only
// OIE a TBM map is scored with, check_cfg) on a dense TBM window, unless SLAMHIP_OPT_TBM_PLANE is off
static bool wants_prob_plane(const slamhip_ctx *ctx, const DeviceMap &m, const slamhip_spe_cfg *cfg) {
  return ctx->tbm_plane && m.cell_model == SLAMHIP_CELL_TBM && m.bytes > 0 && cfg->oope == SLAMHIP_OOPE_OBSTACLE &&
         cfg->oie == SLAMHIP_OIE_DISCREPANCY;
}

static int fill_args(slamhip_ctx *ctx, DeviceMap &m, const slamhip_spe_cfg *cfg, int n_poses,
                     const double *d_poses, const double *d_pose_sc, double *d_scores,
                     ScoreArgs *a) {
  if (ctx->scan_n <= 0) {
    g_last_error = "no scan uploaded";
    return SLAMHIP_ERR_STATE;
  }
  std::memset(a, 0, sizeof(*a));
  a->map.payload = m.d_payload;
  a->map.width = m.width;
  a->map.height = m.height;
  a->map.pitch = m.pitch;
  a->map.origin_x = m.origin_x;
  a->map.origin_y = m.origin_y;
  a->map.scale = m.scale;
  a->map.inv_scale = 1.0 / m.scale;
  for (int k = 0; k < 4; ++k) a->map.unknown[k] = m.unknown[k];
  if (cfg->oope == SLAMHIP_OOPE_GMAPPING && m.cell_model == SLAMHIP_CELL_GMAPPING && m.bytes > 0 && cfg->gm_window == 1 &&
      m.width >= 3 && m.height >= 3) {  // (bytes: a dense window -- the view of a tile pool has none)
    // (the window holds the masks of ONE threshold.  A second scorer configuration with another threshold
    // on the same map does not re-derive them -- a pass over the whole window and a stream stall per alternating call --:
    // it takes the nine-cell form for its calls; the masks follow a threshold that stays, i.e. that asks twice in a row)
    if (m.nbr_ok && m.nbr_th != cfg->gm_fullness_th && m.nbr_other_th != cfg->gm_fullness_th) {
      m.nbr_other_th = cfg->gm_fullness_th;  // first call with this threshold: served without masks
    } else {
      const int rc = map_nbr_masks(ctx, m, cfg->gm_fullness_th);
      if (rc) return rc;
      m.nbr_other_th = m.nbr_th;
      a->map.nbr_ok = 1;
    }
  } else if (cfg->oope == SLAMHIP_OOPE_GMAPPING && m.cell_model == SLAMHIP_CELL_GMAPPING && m.bytes == 0 && cfg->gm_window == 1 &&
             m.nbr_ok && m.nbr_th == cfg->gm_fullness_th) {
    a->map.nbr_ok = 1;  // a tile pool whose owner derived the masks (tile_pool_nbr_masks)
  }
  const size_t c = ctx->scan_stride;
  a->scan.range = ctx->scan_ptr;
  a->scan.cos_a = ctx->scan_ptr + c;
  a->scan.sin_a = ctx->scan_ptr + 2 * c;
  a->scan.weight = ctx->scan_ptr + 3 * c;
  a->scan.factor = ctx->scan_ptr + 4 * c;
  a->scan.n = ctx->scan_n;
  a->scan.tot_w = ctx->scan_tot_w;
  a->poses = d_poses;
  a->pose_sc = d_pose_sc;
  a->scores = d_scores;
  a->n_poses = n_poses;
  a->poses_per_block = 0;
  a->oie = cfg->oie;
  for (int k = 0; k < 4; ++k) a->area[k] = cfg->area[k];
  a->gm.fullness_th = cfg->gm_fullness_th;
  a->gm.window = cfg->gm_window;
  a->gm_info = nullptr;
  a->terms = nullptr;
  a->fprints = nullptr;
  a->model_override = -1;
  if (wants_prob_plane(ctx, m, cfg)) {
    // the per-beam value is a pure function of the cell: gathered from the plane, the map scores like an OCC map
    // whose cells ARE the probabilities (OccupancyOIE returns the cell's value as it is, score_device.h)
    const int rc = map_prob_plane(ctx, m);
    if (rc) return rc;
    a->map.payload = m.d_prob;
    a->map.unknown[0] = tbm_discrepancy_probability(m.unknown[0], m.unknown[1], m.unknown[2], m.unknown[3]);
    a->oie = SLAMHIP_OIE_OCCUPANCY;
    a->model_override = SLAMHIP_CELL_OCC;
  }
  if (cfg->sum_order == SLAMHIP_SUM_SEQUENTIAL && cfg->oope != SLAMHIP_OOPE_GMAPPING) {
    const size_t need = (size_t)n_poses * ctx->scan_n;
    if (need > ctx->terms_cap) {
      SLAMHIP_CHECK(hipStreamSynchronize(ctx->stream));
      if (ctx->stream_b) SLAMHIP_CHECK(hipStreamSynchronize(ctx->stream_b));
      if (ctx->d_terms) hipFree(ctx->d_terms);
      ctx->d_terms = nullptr;
      ctx->terms_cap = 0;
      SLAMHIP_CHECK(hipMalloc(&ctx->d_terms, sizeof(double) * need));
      ctx->terms_cap = need;
    }
    a->terms = ctx->d_terms;
  }
  return SLAMHIP_OK;
}

static int launch_timed(slamhip_ctx *ctx, const ScoreArgs &a, const DeviceMap &m,
                        const slamhip_spe_cfg *cfg, hipStream_t stream) {
  const int order = cfg->oope == SLAMHIP_OOPE_GMAPPING ? SLAMHIP_SUM_TREE256 : cfg->sum_order;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (ctx->profile) {
    // event pairs are only RECORDED here; elapsed times are read in slamhip_profile_read so the
    // timed path never waits on an event
    if (ctx->ev_used + 2 > ctx->ev_pool.size()) {
      const size_t old_n = ctx->ev_pool.size();
      ctx->ev_pool.resize(old_n + 512, nullptr);
      for (size_t i = old_n; i < ctx->ev_pool.size(); ++i) SLAMHIP_CHECK(hipEventCreate(&ctx->ev_pool[i]));
    }
    ctx->ev_kind.resize(ctx->ev_used / 2 + 1, 0);
    ctx->ev_kind[ctx->ev_used / 2] = 0;
    e0 = ctx->ev_pool[ctx->ev_used++];
    e1 = ctx->ev_pool[ctx->ev_used++];
  }
  // one kernel: the events ride on the dispatch (kernel begin..end, as rocprofv3 sees it);
  // strict order is two kernels and is bracketed by recorded events instead
  const bool bracket = ctx->profile && order == SLAMHIP_SUM_SEQUENTIAL;
  if (bracket) SLAMHIP_CHECK(hipEventRecord(e0, stream));
  SLAMHIP_CHECK(launch_score(a, a.model_override >= 0 ? a.model_override : m.cell_model, cfg->oope, order, stream,
                             bracket ? nullptr : e0, bracket ? nullptr : e1));
  if (ctx->profile) {
    if (bracket) SLAMHIP_CHECK(hipEventRecord(e1, stream));
    ctx->prof_launches += 1;
    ctx->prof_units += (long long)a.n_poses * a.scan.n;
  }
  return SLAMHIP_OK;
}

// views of the bound map and the uploaded scan for kernels that are not launched through launch_score
// (the hill-climbing chain); same checks as a scoring call
// the kernels see a tile pool as "payload" and the tiles-per-row count as "pitch" (GMapping OOPE only)
static void tiled_device_map(const TiledTarget *tiled, DeviceMap *v) {
  v->bound = true;
  v->cell_model = SLAMHIP_CELL_GMAPPING;
  v->width = tiled->width;
  v->height = tiled->height;
  v->pitch = tiled->tiles_x;
  v->origin_x = tiled->origin_x;
  v->origin_y = tiled->origin_y;
  v->scale = tiled->scale;
  for (int k = 0; k < 4; ++k) v->unknown[k] = tiled->unknown[k];
  v->d_payload = const_cast<double *>(tiled->pool);
  v->nbr_ok = tiled->nbr_ok != 0;
  v->nbr_th = tiled->nbr_th;
}

int score_views(slamhip_ctx *ctx, int map_id, const slamhip_spe_cfg *cfg, MapView *map, ScanView *scan,
                int *cell_model, const TiledTarget *tiled, int *oie_eff) {
  DeviceMap tiled_view;
  DeviceMap *m = get_map(ctx, map_id);
  if (tiled) {
    if (!cfg || cfg->oope != SLAMHIP_OOPE_GMAPPING) return invalid("per-particle maps are scored by the GMapping kernel only");
    tiled_device_map(tiled, &tiled_view);
    m = &tiled_view;
  }
  if (!m) return invalid("unknown map id");
  int rc = check_cfg(*m, cfg);
  if (rc) return rc;
  ScoreArgs a;
  const bool plane_on = ctx->tbm_plane;
  if (!oie_eff) ctx->tbm_plane = false;  // (a caller that cannot take the substituted OIE gets the map's own view)
  rc = fill_args(ctx, *m, cfg, 1, nullptr, nullptr, nullptr, &a);
  ctx->tbm_plane = plane_on;
  if (rc) return rc;
  *map = a.map;
  *scan = a.scan;
  *cell_model = a.model_override >= 0 ? a.model_override : m->cell_model;
  if (oie_eff) *oie_eff = a.oie;
  return SLAMHIP_OK;
}

// an event pair of the profiling pool (resolved in slamhip_profile_read); null, null when profiling is off
int profile_event_pair(slamhip_ctx *ctx, hipEvent_t *e0, hipEvent_t *e1, int kind) {
  *e0 = *e1 = nullptr;
  if (!ctx->profile) return SLAMHIP_OK;
  ctx->ev_kind.resize(ctx->ev_used / 2 + 1, 0);
  ctx->ev_kind[ctx->ev_used / 2] = (char)kind;
  if (ctx->ev_used + 2 > ctx->ev_pool.size()) {
    const size_t old_n = ctx->ev_pool.size();
    ctx->ev_pool.resize(old_n + 512, nullptr);
    for (size_t i = old_n; i < ctx->ev_pool.size(); ++i) SLAMHIP_CHECK(hipEventCreate(&ctx->ev_pool[i]));
  }
  *e0 = ctx->ev_pool[ctx->ev_used++];
  *e1 = ctx->ev_pool[ctx->ev_used++];
  return SLAMHIP_OK;
}

int ProfilePairGuard::open(slamhip_ctx *ctx, hipStream_t st, int kind) {
  stream = st;
  const int rc = profile_event_pair(ctx, &e0, &e1, kind);
  if (rc) return rc;
  if (e0) {
    SLAMHIP_CHECK(hipEventRecord(e0, st));
    closed = false;
  }
  return SLAMHIP_OK;
}
int ProfilePairGuard::close() {
  if (closed || !e1) return SLAMHIP_OK;
  closed = true;
  SLAMHIP_CHECK(hipEventRecord(e1, stream));
  return SLAMHIP_OK;
}

static int profile_resolve(slamhip_ctx *ctx) {
  if (ctx->ev_used == 0) return SLAMHIP_OK;
  // whatever happens below, the pool starts afresh: a failure here must not wedge every later read
  const size_t used = ctx->ev_used;
  ctx->ev_used = 0;
  std::vector<char> kinds;
  kinds.swap(ctx->ev_kind);
  SLAMHIP_CHECK(hipStreamSynchronize(ctx->stream));
  if (ctx->stream_b) SLAMHIP_CHECK(hipStreamSynchronize(ctx->stream_b));
  for (size_t i = 0; i + 1 < used; i += 2) {
    float ms = 0.f;
    // a pair whose end was never recorded (a launch that failed between the two) is skipped, not fatal
    if (hipEventElapsedTime(&ms, ctx->ev_pool[i], ctx->ev_pool[i + 1]) != hipSuccess) {
      (void)hipGetLastError();
      continue;
    }
    if (i / 2 < kinds.size() && kinds[i / 2] == 1) ctx->prof_k6_ms += ms;
    else ctx->prof_ms += ms;
  }
  return SLAMHIP_OK;
}

// GMapping OOPE cache carried ACROSS poses (Q19): the kernel resolves runs inside each pose; a
// pose whose first beam lands in the cell the previous call ended in re-uses that cached value
// for its whole first run.  Applied in call order on the host.
static void gm_carry_fixup(slamhip_ctx *ctx, int n_poses) {
  if (ctx->gm_exact_last) return;  // (k_score_gmapping_exact applied the cache itself, pose after pose)
  const double tot_w = ctx->scan_tot_w;
  for (int p = 0; p < n_poses; ++p) {
    GmPoseInfo &gi = ctx->h_gm_info[p];
    double last_v = gi.last_v;
    if (ctx->gm_prob != -1.0 && gi.first_cx == ctx->gm_cx && gi.first_cy == ctx->gm_cy) {
      const double c = ctx->gm_prob;
      if (c != gi.v0) {
        double delta = 0.0;
        for (int b = 0; b < gi.run0_len; ++b)
          delta += (c * ctx->h_weight[b]) * ctx->h_factor[b] - (gi.v0 * ctx->h_weight[b]) * ctx->h_factor[b];
        if (tot_w != 0.0) ctx->h_scores[p] += delta / tot_w;
      }
      if (gi.last_head == 0) last_v = c;
    }
    ctx->gm_cx = gi.last_cx;
    ctx->gm_cy = gi.last_cy;
    ctx->gm_prob = last_v;
  }
}

int lane_fork(slamhip_ctx *ctx) {
  if (!ctx->stream_b) {
    SLAMHIP_CHECK(hipStreamCreateWithFlags(&ctx->stream_b, hipStreamNonBlocking));
    SLAMHIP_CHECK(hipHostMalloc(&ctx->h_done_flag_b, sizeof(unsigned), kPinned));
    *ctx->h_done_flag_b = 0;
    SLAMHIP_CHECK(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
  }
  SLAMHIP_CHECK(hipEventRecord(ctx->ev_fork, ctx->stream));
  SLAMHIP_CHECK(hipStreamWaitEvent(ctx->stream_b, ctx->ev_fork, 0));
  return SLAMHIP_OK;
}

int score_wait(slamhip_ctx *ctx, unsigned seq, int lane) {
  if (seq == 0) return SLAMHIP_OK;  // the launch was synchronous
  volatile unsigned *flag = lane ? ctx->h_done_flag_b : ctx->h_done_flag;
  const hipStream_t stream = lane ? ctx->stream_b : ctx->stream;
  unsigned long long spins = 0;
  // launches publish increasing numbers on one stream: "reached seq" = the signed distance is >= 0
  while ((int)(*flag - seq) < 0) {
    __builtin_ia32_pause();
    if ((++spins & 0xfffffull) == 0) {
      // ~every few ms: make sure the launch did not fail asynchronously
      hipError_t q = hipStreamQuery(stream);
      if (q != hipSuccess && q != hipErrorNotReady) return hip_fail(q, "scoring kernel");
      if (q == hipSuccess && (int)(*flag - seq) < 0) {
        set_error("scoring kernel finished without publishing its completion flag");
        return SLAMHIP_ERR_HIP;
      }
    }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  return SLAMHIP_OK;
}

// Which build of glibc's sin / cos / exp the host's libm runs (libm_exact.h): the two restated builds are evaluated
// on pseudo-random arguments until each function has eight where they differ, and libm -- called through volatile
// pointers, sin and cos one at a time (the compiler would fuse a pair into sincos(), which has no FMA build) -- has to
// agree with one of them on all of those.  Process-wide, computed once.
int libm_variant() {
  static const int variant = [] {
    double (*volatile p_sin)(double) = ::sin;
    double (*volatile p_cos)(double) = ::cos;
    double (*volatile p_exp)(double) = ::exp;
    unsigned long long st = 0x243f6a8885a308d3ull;
    auto next = [&st]() {
      st += 0x9e3779b97f4a7c15ull;
      unsigned long long z = st;
      z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
      z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
      return (double)((z ^ (z >> 31)) >> 11) * 0x1p-53;
    };
    int votes_fma = 0, votes_plain = 0, probes = 0;
    int found[3] = {0, 0, 0};
    for (long it = 0; it < 4000000 && (found[0] < 8 || found[1] < 8 || found[2] < 8); ++it) {
      const double xt = 0.2 + 6.0 * next(), xe = -2.0 * next();
      for (int f = 0; f < 3; ++f) {
        if (found[f] >= 8) continue;
        const double x = f == 2 ? xe : xt;
        const double a = f == 0 ? libm_exact::sin_<true>(x) : (f == 1 ? libm_exact::cos_<true>(x) : libm_exact::exp_<true>(x));
        const double b = f == 0 ? libm_exact::sin_<false>(x) : (f == 1 ? libm_exact::cos_<false>(x) : libm_exact::exp_<false>(x));
        if (a == b) continue;
        const double l = f == 0 ? p_sin(x) : (f == 1 ? p_cos(x) : p_exp(x));
        ++found[f];
        ++probes;
        votes_fma += l == a;
        votes_plain += l == b;
      }
    }
    if (probes < 24) return -1;
    if (votes_fma == probes) return 1;
    if (votes_plain == probes) return 0;
    return -1;
  }();
  return variant;
}

// the beam angles of the current scan in HBM (uploaded on first use)
static int exact_angles(slamhip_ctx *ctx, const double **d_angle) {
  const int n = ctx->scan_n;
  if ((int)ctx->h_scan_angle.size() != n) {
    set_error("SLAMHIP_POSE_TRIG_RAW_EXACT needs the angles of the current scan's points: slamhip_scan_set_angles (or "
              "slamhip_scan_filter_upload) after the scan upload");
    return SLAMHIP_ERR_STATE;
  }
  if (!ctx->scan_angle_on_device) {
    if (n > ctx->scan_angle_cap) {
      SLAMHIP_CHECK(hipStreamSynchronize(ctx->stream));
      if (ctx->d_scan_angle) hipFree(ctx->d_scan_angle);
      ctx->d_scan_angle = nullptr;
      ctx->scan_angle_cap = 0;
      const int cap = (n + 2047) & ~2047;
      SLAMHIP_CHECK(hipMalloc(&ctx->d_scan_angle, sizeof(double) * cap));
      ctx->scan_angle_cap = cap;
    }
    SLAMHIP_CHECK(hipStreamSynchronize(ctx->stream));  // (an earlier exact call may still read the old angles)
    SLAMHIP_CHECK(hipMemcpy(ctx->d_scan_angle, ctx->h_scan_angle.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    ctx->scan_angle_on_device = true;
  }
  *d_angle = ctx->d_scan_angle;
  return SLAMHIP_OK;
}

// The libm-exact modes of a staged batch (exact_kernels.hip), synchronous: results are in h_scores on return.
//   * GMapping OOPE + (SLAMHIP_SUM_SEQUENTIAL or RAW_EXACT): ONE launch of k_score_gmapping_exact walks the poses in
//     order with the reference's cache on the device (ctx->gm_* in, ctx->gm_* out: no host fix-up afterwards);
//   * RAW_EXACT over the other OOPEs: cos / sin(theta_p + a_b) tabulated for the batch, then the ordinary scoring
//     kernels pose by pose over the pose's own table with the identity as pose rotation.
static int score_exact(slamhip_ctx *ctx, DeviceMap &m, const slamhip_spe_cfg *cfg, int n_poses, const TiledTarget *tiled,
                       int off) {
  const int variant = libm_variant();
  if (variant < 0) {
    set_error("this host's libm matches neither build of glibc's sin / cos / exp restated in csrc/libm_exact.h: the exact "
              "modes are not available here");
    return SLAMHIP_ERR_UNSUPPORTED;
  }
  const bool fma = variant == 1;
  const bool raw = cfg->pose_trig == SLAMHIP_POSE_TRIG_RAW_EXACT;
  const bool gm = cfg->oope == SLAMHIP_OOPE_GMAPPING;
  const double *d_angle = nullptr;
  if (raw) {
    const int rc = exact_angles(ctx, &d_angle);
    if (rc) return rc;
  }
  const double *poses_src = ctx->h_poses + 3 * (size_t)off;
  const double *sc_src = cfg->pose_trig == SLAMHIP_POSE_TRIG_HOST ? ctx->h_pose_sc + 2 * (size_t)off : nullptr;
  ScoreArgs a;
  int rc = fill_args(ctx, m, cfg, n_poses, poses_src, sc_src, ctx->h_scores + off, &a);
  if (rc) return rc;
  const int n = ctx->scan_n;
  if (gm) {
    if (tiled) {
      a.tables = tiled->tables;
      a.pose_slot = ctx->h_pose_slot + off;
      a.table_stride = tiled->table_stride;
    }
    if (!ctx->d_gm_exact_cache) SLAMHIP_CHECK(hipMalloc(&ctx->d_gm_exact_cache, 16));
    struct {
      int cx, cy;
      double prob;
    } c = {ctx->gm_cx, ctx->gm_cy, ctx->gm_prob};
    SLAMHIP_CHECK(hipMemcpyAsync(ctx->d_gm_exact_cache, &c, sizeof c, hipMemcpyHostToDevice, ctx->stream));
    hipError_t e = launch_score_gmapping_exact(fma, a, d_angle, raw ? 1 : 0, ctx->d_gm_exact_cache, ctx->stream);
    if (e != hipSuccess) {
      if (e == hipErrorInvalidValue) return invalid("the exact GMapping kernel holds at most 3840 filtered beams per scan");
      return hip_fail(e, "k_score_gmapping_exact");
    }
    SLAMHIP_CHECK(hipMemcpyAsync(&c, ctx->d_gm_exact_cache, sizeof c, hipMemcpyDeviceToHost, ctx->stream));
    SLAMHIP_CHECK(hipStreamSynchronize(ctx->stream));
    ctx->gm_cx = c.cx;
    ctx->gm_cy = c.cy;
    ctx->gm_prob = c.prob;
    ctx->gm_exact_last = true;
    // (no per-pose side outputs: a caller that replays them finds "no carry-in, nothing to fix")
    for (int p = 0; p < n_poses; ++p) std::memset(&ctx->h_gm_info[off + p], 0, sizeof(GmPoseInfo));
    if (ctx->profile) {
      ctx->prof_launches += 1;
      ctx->prof_units += (long long)n_poses * n;
    }
    return SLAMHIP_OK;
  }
  const size_t stride = (size_t)((n + 7) & ~7);
  const size_t need = 2 * stride * (size_t)n_poses + 2 * (size_t)n_poses;
  if (need > ctx->exact_trig_cap) {
    SLAMHIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->d_exact_trig) hipFree(ctx->d_exact_trig);
    ctx->d_exact_trig = nullptr;
    ctx->exact_trig_cap = 0;
    SLAMHIP_CHECK(hipMalloc(&ctx->d_exact_trig, sizeof(double) * need));
    ctx->exact_trig_cap = need;
  }
  double *d_cos = ctx->d_exact_trig, *d_sin = d_cos + stride * (size_t)n_poses, *d_id = d_sin + stride * (size_t)n_poses;
  SLAMHIP_CHECK(launch_exact_beam_trig(fma, poses_src, n_poses, d_angle, n, stride, d_cos, d_sin, d_id, ctx->stream));
  if (ctx->want_fprints && cfg->sum_order == SLAMHIP_SUM_TREE256) a.fprints = ctx->h_fprints + off;
  for (int p = 0; p < n_poses; ++p) {
    ScoreArgs ap = a;
    ap.n_poses = 1;
    ap.poses = a.poses + 3 * (size_t)p;
    ap.pose_sc = d_id + 2 * (size_t)p;
    ap.scores = a.scores + p;
    ap.scan.cos_a = d_cos + stride * (size_t)p;
    ap.scan.sin_a = d_sin + stride * (size_t)p;
    if (a.terms) ap.terms = a.terms + (size_t)p * n;
    if (a.fprints) ap.fprints = a.fprints + p;
    SLAMHIP_CHECK(launch_score(ap, a.model_override >= 0 ? a.model_override : m.cell_model, cfg->oope, cfg->sum_order,
                               ctx->stream));
  }
  SLAMHIP_CHECK(hipStreamSynchronize(ctx->stream));
  if (ctx->profile) {
    ctx->prof_launches += n_poses;
    ctx->prof_units += (long long)n_poses * n;
  }
  return SLAMHIP_OK;
}

int score_staged(slamhip_ctx *ctx, int map_id, const slamhip_spe_cfg *cfg, int n_poses,
                 const TiledTarget *tiled, int off, unsigned *async_seq, int lane) {
  if (async_seq) *async_seq = 0;
  if (lane && (!ctx->low_latency || ctx->stage_poses || !ctx->stream_b))
    return invalid("the second launch lane needs the zero-copy path and lane_fork()");
  DeviceMap tiled_view;
  DeviceMap *m = nullptr;
  if (tiled) {
    if (!cfg || cfg->oope != SLAMHIP_OOPE_GMAPPING || !ctx->low_latency)
      return invalid("per-particle maps are scored by the GMapping kernel on the zero-copy path only");
    tiled_device_map(tiled, &tiled_view);
    m = &tiled_view;
  } else {
    m = get_map(ctx, map_id);
  }
  if (!m) return invalid("unknown map id");
  int rc = check_cfg(*m, cfg);
  if (rc) return rc;
  if (n_poses <= 0) return SLAMHIP_OK;
  const bool host_trig = cfg->pose_trig == SLAMHIP_POSE_TRIG_HOST;
  if (cfg->pose_trig != SLAMHIP_POSE_TRIG_DEVICE && cfg->pose_trig != SLAMHIP_POSE_TRIG_HOST &&
      cfg->pose_trig != SLAMHIP_POSE_TRIG_RAW_EXACT)
    return invalid("unknown pose_trig");
  if (off < 0 || off + n_poses > ctx->pose_cap) return invalid("staging window outside the pose capacity");
  ctx->gm_exact_last = false;
  if (off != 0 && (!ctx->low_latency || ctx->stage_poses))
    return invalid("staging windows need the zero-copy path");
  if (host_trig) {
    for (int p = off; p < off + n_poses; ++p) {
      // one sincos call per pose: the reference build (g++ -O3) fuses the sin/cos pair of
      // set_base_angle (trigonometry_utils.h:57-60) into glibc's sincos, whose last bit can
      // differ from separate sin()/cos() calls
      ::sincos(ctx->h_poses[3 * p + 2], &ctx->h_pose_sc[2 * p], &ctx->h_pose_sc[2 * p + 1]);
    }
  }
  const bool gm = cfg->oope == SLAMHIP_OOPE_GMAPPING;
  // the libm-exact modes (synchronous; the host-driven callers find *async_seq == 0 = "nothing to wait for")
  if (cfg->pose_trig == SLAMHIP_POSE_TRIG_RAW_EXACT || (gm && cfg->sum_order == SLAMHIP_SUM_SEQUENTIAL)) {
    if (lane) return invalid("the exact modes run on the context's first launch lane");
    return score_exact(ctx, *m, cfg, n_poses, tiled, off);
  }
  ScoreArgs a;
  if (ctx->low_latency) {
    // zero-copy: the kernel reads poses from / writes scores to the pinned staging buffers; a
    // 1-thread kernel behind it publishes the launch number in pinned memory; the host spins.
    // `off` selects a window of the staging buffers, so two launches can be in flight (the filter
    // plans one half of its particles while the GPU scores the other).
    const double *poses_src = ctx->h_poses + 3 * (size_t)off;
    const double *sc_src = host_trig ? ctx->h_pose_sc + 2 * (size_t)off : nullptr;
    if (ctx->stage_poses) {
      SLAMHIP_CHECK(hipMemcpyAsync(ctx->d_poses, ctx->h_poses, sizeof(double) * 3 * n_poses,
                                   hipMemcpyHostToDevice, ctx->stream));
      poses_src = ctx->d_poses;
      if (host_trig) {
        SLAMHIP_CHECK(hipMemcpyAsync(ctx->d_pose_sc, ctx->h_pose_sc, sizeof(double) * 2 * n_poses,
                                     hipMemcpyHostToDevice, ctx->stream));
        sc_src = ctx->d_pose_sc;
      }
    }
    rc = fill_args(ctx, *m, cfg, n_poses, poses_src, sc_src, ctx->h_scores + off, &a);
    if (rc) return rc;
    if (gm) a.gm_info = ctx->h_gm_info + off;
    if (ctx->want_fprints && cfg->oope != SLAMHIP_OOPE_GMAPPING && cfg->sum_order == SLAMHIP_SUM_TREE256)
      a.fprints = ctx->h_fprints + off;
    if (tiled) {
      a.tables = tiled->tables;
      a.pose_slot = ctx->h_pose_slot + off;
      a.table_stride = tiled->table_stride;
    }
    unsigned &counter = lane ? ctx->seq_b : ctx->seq;
    unsigned seq = ++counter;
    if (seq == 0) seq = ++counter;
    const hipStream_t stream = lane ? ctx->stream_b : ctx->stream;
    rc = launch_timed(ctx, a, *m, cfg, stream);
    if (rc) return rc;
    SLAMHIP_CHECK(launch_publish(lane ? ctx->h_done_flag_b : ctx->h_done_flag, seq, stream));
    if (async_seq) {
      *async_seq = seq;
      return SLAMHIP_OK;
    }
    return score_wait(ctx, seq, lane);
  }
  if (host_trig)
    SLAMHIP_CHECK(hipMemcpyAsync(ctx->d_pose_sc, ctx->h_pose_sc, sizeof(double) * 2 * n_poses,
                                 hipMemcpyHostToDevice, ctx->stream));
  SLAMHIP_CHECK(hipMemcpyAsync(ctx->d_poses, ctx->h_poses, sizeof(double) * 3 * n_poses,
                               hipMemcpyHostToDevice, ctx->stream));
  rc = fill_args(ctx, *m, cfg, n_poses, ctx->d_poses, host_trig ? ctx->d_pose_sc : nullptr,
                 ctx->d_scores, &a);
  if (rc) return rc;
  if (gm) a.gm_info = ctx->d_gm_info;
  rc = launch_timed(ctx, a, *m, cfg, ctx->stream);
  if (rc) return rc;
  SLAMHIP_CHECK(hipMemcpyAsync(ctx->h_scores, ctx->d_scores, sizeof(double) * n_poses,
                               hipMemcpyDeviceToHost, ctx->stream));
  if (gm)
    SLAMHIP_CHECK(hipMemcpyAsync(ctx->h_gm_info, ctx->d_gm_info, sizeof(GmPoseInfo) * n_poses,
                                 hipMemcpyDeviceToHost, ctx->stream));
  SLAMHIP_CHECK(hipStreamSynchronize(ctx->stream));
  return SLAMHIP_OK;
}

}  // namespace slamhip

using namespace slamhip;

extern "C" {

const char *slamhip_last_error(void) { return g_last_error.c_str(); }

int slamhip_device_count(int *count) {
  if (!count) return invalid("null count");
  *count = 0;
  hipError_t e = hipGetDeviceCount(count);
  if (e != hipSuccess) {
    *count = 0;
    return hip_fail(e, "hipGetDeviceCount");
  }
  return SLAMHIP_OK;
}

int slamhip_ctx_create(int device, slamhip_ctx **out) {
  if (!out) return invalid("null out");
  *out = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    g_last_error = "no HIP device available (the scorer has no CPU fallback)";
    return SLAMHIP_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= count) return invalid("device index out of range");
  SLAMHIP_CHECK(hipSetDevice(device));
  auto *ctx = new slamhip_ctx;
  ctx->device = device;
  e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete ctx;
    return hip_fail(e, "hipStreamCreate");
  }
  e = hipHostMalloc(&ctx->h_done_flag, sizeof(unsigned), kPinned);
  if (e != hipSuccess) {
    hipStreamDestroy(ctx->stream);
    delete ctx;
    return hip_fail(e, "completion flag allocation");
  }
  *ctx->h_done_flag = 0;
  *out = ctx;
  return SLAMHIP_OK;
}

int slamhip_ctx_set_option(slamhip_ctx *ctx, int option, int value) {
  if (!ctx) return invalid("null argument");
  switch (option) {
    case SLAMHIP_OPT_LOW_LATENCY: ctx->low_latency = value != 0; break;
    case SLAMHIP_OPT_STAGE_POSES: ctx->stage_poses = value != 0; break;
    case SLAMHIP_OPT_FILTER_CHAINS: ctx->filter_chains = value != 0; break;
    case SLAMHIP_OPT_K6_PATH:
      if (value < 0 || value > 2) return invalid("SLAMHIP_OPT_K6_PATH: 0 (default), 1 (counting sort) or 2 (radix sort)");
      ctx->k6_path = value;
      break;
    case SLAMHIP_OPT_K6_BATCH_FAST: ctx->k6_batch_fast = value != 0; break;
    case SLAMHIP_OPT_K6_BATCH_KEY64: ctx->k6_batch_key64 = value != 0; break;
    case SLAMHIP_OPT_RESIDENT_CHAINS: ctx->resident_chains = value != 0; break;
    case SLAMHIP_OPT_TBM_PLANE: ctx->tbm_plane = value != 0; break;
    case SLAMHIP_OPT_INERT_TAIL:
      if (value < 0 || value > 2) return invalid("SLAMHIP_OPT_INERT_TAIL: 0 (off), 1 (identical poses) or 2 (default: certified poses too)");
      ctx->inert_tail = value;
      break;
    default: return invalid("unknown option");
  }
  return SLAMHIP_OK;
}

int slamhip_ctx_get_option(slamhip_ctx *ctx, int option, int *value) {
  if (!ctx || !value) return invalid("null argument");
  switch (option) {
    case SLAMHIP_OPT_LOW_LATENCY: *value = ctx->low_latency; break;
    case SLAMHIP_OPT_STAGE_POSES: *value = ctx->stage_poses; break;
    case SLAMHIP_OPT_FILTER_CHAINS: *value = ctx->filter_chains; break;
    case SLAMHIP_OPT_K6_PATH: *value = ctx->k6_path; break;
    case SLAMHIP_OPT_K6_BATCH_FAST: *value = ctx->k6_batch_fast; break;
    case SLAMHIP_OPT_K6_BATCH_KEY64: *value = ctx->k6_batch_key64; break;
    case SLAMHIP_OPT_RESIDENT_CHAINS: *value = ctx->resident_chains; break;
    case SLAMHIP_OPT_TBM_PLANE: *value = ctx->tbm_plane; break;
    case SLAMHIP_OPT_INERT_TAIL: *value = ctx->inert_tail; break;
    default: return invalid("unknown option");
  }
  return SLAMHIP_OK;
}

int slamhip_ctx_destroy(slamhip_ctx *ctx) {
  if (!ctx) return SLAMHIP_OK;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  for (auto &m : ctx->maps) {
    if (m.d_payload) hipFree(m.d_payload);
    if (m.d_aux) hipFree(m.d_aux);
    if (m.d_prob) hipFree(m.d_prob);
  }
  if (ctx->d_scan) hipFree(ctx->d_scan);
  if (ctx->d_scan_angle) hipFree(ctx->d_scan_angle);
  if (ctx->d_exact_trig) hipFree(ctx->d_exact_trig);
  if (ctx->d_gm_exact_cache) hipFree(ctx->d_gm_exact_cache);
  for (auto &sl : ctx->scan_slots)
    if (sl.d) hipFree(sl.d);
  for (int k = 0; k < 2; ++k) {
    if (ctx->h_scan_stage[k]) hipHostFree(ctx->h_scan_stage[k]);
    if (ctx->scan_stage_done[k]) hipEventDestroy(ctx->scan_stage_done[k]);
    if (k == 0 && ctx->h_scan_pulled) hipHostFree(ctx->h_scan_pulled);
    if (k == 0 && ctx->d_scan_pull_count) hipFree(ctx->d_scan_pull_count);
  }
  if (ctx->d_poses) hipFree(ctx->d_poses);
  if (ctx->d_scores) hipFree(ctx->d_scores);
  if (ctx->d_pose_sc) hipFree(ctx->d_pose_sc);
  if (ctx->d_gm_info) hipFree(ctx->d_gm_info);
  if (ctx->h_poses) hipHostFree(ctx->h_poses);
  if (ctx->h_scores) hipHostFree(ctx->h_scores);
  if (ctx->h_pose_sc) hipHostFree(ctx->h_pose_sc);
  if (ctx->h_gm_info) hipHostFree(ctx->h_gm_info);
  if (ctx->h_fprints) hipHostFree(ctx->h_fprints);
  if (ctx->h_pose_slot) hipHostFree(ctx->h_pose_slot);
  if (ctx->d_terms) hipFree(ctx->d_terms);
  if (ctx->d_dirty_xy) hipFree(ctx->d_dirty_xy);
  if (ctx->d_dirty_val) hipFree(ctx->d_dirty_val);
  for (hipEvent_t ev : ctx->ev_pool)
    if (ev) hipEventDestroy(ev);
  if (ctx->h_done_flag) hipHostFree(ctx->h_done_flag);
  if (ctx->stream_b) {
    hipStreamSynchronize(ctx->stream_b);
    hipStreamDestroy(ctx->stream_b);
    hipHostFree(ctx->h_done_flag_b);
    hipEventDestroy(ctx->ev_fork);
  }
  mu_release(ctx);
  shard_release(ctx);
  hipStreamDestroy(ctx->stream);
  delete ctx;
  return SLAMHIP_OK;
}

int slamhip_ctx_synchronize(slamhip_ctx *ctx) {
  if (!ctx) return invalid("null ctx");
  SLAMHIP_CHECK(hipStreamSynchronize(ctx->stream));
  return SLAMHIP_OK;
}

void *slamhip_ctx_stream(slamhip_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

// ---------------------------------------------------------------------------------- map mirror
int slamhip_map_bind(slamhip_ctx *ctx, int map_id, int cell_model, int width, int height,
                     int origin_x, int origin_y, double scale, const double *unknown_payload) {
  if (!ctx) return invalid("null ctx");
  if (map_id < 0 || map_id > 4095) return invalid("map_id out of range [0, 4095]");
  if (cell_model < SLAMHIP_CELL_OCC || cell_model > SLAMHIP_CELL_GMAPPING)
    return invalid("unknown cell model");
  if (width <= 0 || height <= 0 || !(scale > 0) || !unknown_payload)
    return invalid("bad map geometry");
  SLAMHIP_CHECK(hipSetDevice(ctx->device));
  if ((int)ctx->maps.size() <= map_id) ctx->maps.resize(map_id + 1);
  DeviceMap old = ctx->maps[map_id];
  DeviceMap nm;
  nm.bound = true;
  nm.cell_model = cell_model;
  nm.width = width;
  nm.height = height;
  nm.pitch = (width + 15) & ~15;
  nm.origin_x = origin_x;
  nm.origin_y = origin_y;
  nm.scale = scale;
  const int sh = cell_stride_host(cell_model), cd = cell_doubles(cell_model);
  for (int k = 0; k < 4; ++k) nm.unknown[k] = k < sh ? unknown_payload[k] : 0.0;
  nm.bytes = (size_t)nm.pitch * height * cd * sizeof(double);
  SLAMHIP_CHECK(hipMalloc(&nm.d_payload, nm.bytes));
  SLAMHIP_CHECK(launch_fill_cells(nm.d_payload, (size_t)nm.pitch * height, cd, nm.unknown, ctx->stream));
  if (old.bound && old.cell_model == cell_model && old.d_payload) {
    // growth of an unbounded map: the old window keeps its EXTERNAL coordinates, i.e. it moves by
    // the origin shift in internal coordinates (plain_grid_map.h:133-173)
    const int dx = origin_x - old.origin_x, dy = origin_y - old.origin_y;
    const int sx0 = std::max(0, -dx), sy0 = std::max(0, -dy);
    const int sx1 = std::min(old.width, width - dx), sy1 = std::min(old.height, height - dy);
    if (sx1 > sx0 && sy1 > sy0) {
      const size_t cb = cd * sizeof(double);
      SLAMHIP_CHECK(hipMemcpy2DAsync(
          nm.d_payload + ((size_t)(sy0 + dy) * nm.pitch + (sx0 + dx)) * cd, nm.pitch * cb,
          old.d_payload + ((size_t)sy0 * old.pitch + sx0) * cd, old.pitch * cb,
          (size_t)(sx1 - sx0) * cb, sy1 - sy0, hipMemcpyDeviceToDevice, ctx->stream));
    }
  }
  if (old.bound && old.cell_model == cell_model && old.d_aux && old.aux_stride) {
    // the update counters move with their cells
    const size_t ab = (size_t)nm.pitch * height * old.aux_stride * sizeof(double);
    SLAMHIP_CHECK(hipMalloc(&nm.d_aux, ab));
    SLAMHIP_CHECK(hipMemsetAsync(nm.d_aux, 0, ab, ctx->stream));
    nm.aux_stride = old.aux_stride;
    const int dx = origin_x - old.origin_x, dy = origin_y - old.origin_y;
    const int sx0 = std::max(0, -dx), sy0 = std::max(0, -dy);
    const int sx1 = std::min(old.width, width - dx), sy1 = std::min(old.height, height - dy);
    if (sx1 > sx0 && sy1 > sy0) {
      const size_t cb = old.aux_stride * sizeof(double);
      SLAMHIP_CHECK(hipMemcpy2DAsync(
          nm.d_aux + ((size_t)(sy0 + dy) * nm.pitch + (sx0 + dx)) * old.aux_stride, nm.pitch * cb,
          old.d_aux + ((size_t)sy0 * old.pitch + sx0) * old.aux_stride, old.pitch * cb,
          (size_t)(sx1 - sx0) * cb, sy1 - sy0, hipMemcpyDeviceToDevice, ctx->stream));
    }
  }
  SLAMHIP_CHECK(hipStreamSynchronize(ctx->stream));
  if (old.d_payload) hipFree(old.d_payload);
  if (old.d_aux) hipFree(old.d_aux);
  if (old.d_prob) hipFree(old.d_prob);  // (the re-bound window has no plane until a scorer asks again)
  if (old.bound && old.cell_model == cell_model) {
    nm.auto_grow = old.auto_grow;
    nm.grown = old.grown;
  }
  ctx->maps[map_id] = nm;
  return SLAMHIP_OK;
}

int slamhip_map_set_auto_grow(slamhip_ctx *ctx, int map_id, int on) {
  DeviceMap *m = get_map(ctx, map_id);
  if (!m) return invalid("unknown map id");
  m->auto_grow = on != 0;
  return SLAMHIP_OK;
}

int slamhip_map_set_deferred(slamhip_ctx *ctx, int on) {
  if (!ctx) return invalid("null context");
  SLAMHIP_CHECK(hipSetDevice(ctx->device));
  if (!on) {  // nothing stays queued behind the caller's back
    long long nu = 0;
    const int rc = slamhip_map_drain(ctx, &nu);
    if (rc) return rc;
  }
  mu_set_deferred(ctx, on != 0);
  return SLAMHIP_OK;
}

int slamhip_map_drain(slamhip_ctx *ctx, long long *n_updates) {
  if (!ctx) return invalid("null context");
  SLAMHIP_CHECK(hipSetDevice(ctx->device));
  long long nu = 0;
  int err = 0;
  const int rc = mu_drain(ctx, &nu, &err);
  if (n_updates) *n_updates = nu;
  if (rc) return rc;
  if (err) {
    set_error(err == 2 ? "internal: the device counted more cell updates than the host sized the buffers for"
                       : "a queued map update met a beam that leaves the bound window (cells inside it were updated)");
    return SLAMHIP_ERR_STATE;
  }
  return SLAMHIP_OK;
}

int slamhip_map_info(slamhip_ctx *ctx, int map_id, int *cell_model, int *width, int *height, int *origin_x,
                     int *origin_y, double *scale, long long *times_grown) {
  DeviceMap *m = get_map(ctx, map_id);
  if (!m) return invalid("unknown map id");
  if (cell_model) *cell_model = m->cell_model;
  if (width) *width = m->width;
  if (height) *height = m->height;
  if (origin_x) *origin_x = m->origin_x;
  if (origin_y) *origin_y = m->origin_y;
  if (scale) *scale = m->scale;
  if (times_grown) *times_grown = m->grown;
  return SLAMHIP_OK;
}

int slamhip_map_release(slamhip_ctx *ctx, int map_id) {
  DeviceMap *m = get_map(ctx, map_id);
  if (!m) return invalid("unknown map id");
  SLAMHIP_CHECK(hipStreamSynchronize(ctx->stream));
  if (m->d_payload) hipFree(m->d_payload);
  if (m->d_aux) hipFree(m->d_aux);
  if (m->d_prob) hipFree(m->d_prob);
  *m = DeviceMap{};
  return SLAMHIP_OK;
}

int slamhip_map_upload_window(slamhip_ctx *ctx, int map_id, int x0, int y0, int w, int h,
                              const double *payload) {
  DeviceMap *m = get_map(ctx, map_id);
  if (!m) return invalid("unknown map id");
  if (!payload || w <= 0 || h <= 0 || x0 < 0 || y0 < 0 || x0 + w > m->width || y0 + h > m->height)
    return invalid("window outside the bound map");
  const int sh = cell_stride_host(m->cell_model), cd = cell_doubles(m->cell_model);
  const size_t bytes = (size_t)w * h * sh * sizeof(double);
  double *d_tmp = nullptr;
  SLAMHIP_CHECK(hipMalloc(&d_tmp, bytes));
  hipError_t e = hipMemcpyAsync(d_tmp, payload, bytes, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess)
    e = launch_repack_window(m->d_payload, m->pitch, cd, d_tmp, sh, x0, y0, w, h, ctx->stream);
  if (e == hipSuccess && m->nbr_ok)  // the masks of the written cells and of the ring around them
    e = launch_nbr_build(m->d_payload, m->width, m->height, m->pitch, m->nbr_th, x0 - 1, y0 - 1, w + 2, h + 2, ctx->stream);
  if (e == hipSuccess && m->prob_ok)  // the probability plane of the written cells
    e = launch_prob_build(m->d_payload, m->d_prob, m->width, m->height, m->pitch, x0, y0, w, h, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  hipFree(d_tmp);
  if (e != hipSuccess) return hip_fail(e, "map_upload_window");
  return SLAMHIP_OK;
}

int slamhip_map_download_window(slamhip_ctx *ctx, int map_id, int x0, int y0, int w, int h,
                                double *out) {
  DeviceMap *m = get_map(ctx, map_id);
  if (!m) return invalid("unknown map id");
  if (!out || w <= 0 || h <= 0 || x0 < 0 || y0 < 0 || x0 + w > m->width || y0 + h > m->height)
    return invalid("window outside the bound map");
  const int sh = cell_stride_host(m->cell_model), cd = cell_doubles(m->cell_model);
  std::vector<double> tmp((size_t)w * h * cd);
  const size_t cb = cd * sizeof(double);
  SLAMHIP_CHECK(hipMemcpy2DAsync(tmp.data(), w * cb, m->d_payload + ((size_t)y0 * m->pitch + x0) * cd,
                                 m->pitch * cb, w * cb, h, hipMemcpyDeviceToHost, ctx->stream));
  SLAMHIP_CHECK(hipStreamSynchronize(ctx->stream));
  for (size_t i = 0; i < (size_t)w * h; ++i)
    for (int k = 0; k < sh; ++k) out[i * sh + k] = tmp[i * cd + k];
  return SLAMHIP_OK;
}

int slamhip_map_apply_dirty(slamhip_ctx *ctx, int map_id, int n, const int *coords_xy,
                            const double *payloads) {
  DeviceMap *m = get_map(ctx, map_id);
  if (!m) return invalid("unknown map id");
  if (n < 0 || (n > 0 && (!coords_xy || !payloads))) return invalid("bad dirty log");
  if (n == 0) return SLAMHIP_OK;
  for (int i = 0; i < n; ++i)
    if (coords_xy[2 * i] < 0 || coords_xy[2 * i] >= m->width || coords_xy[2 * i + 1] < 0 ||
        coords_xy[2 * i + 1] >= m->height)
      return invalid("dirty cell outside the bound window (re-bind the grown map first)");
  const int sh = cell_stride_host(m->cell_model), cd = cell_doubles(m->cell_model);
  if (n > ctx->dirty_cap) {
    SLAMHIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->d_dirty_xy) hipFree(ctx->d_dirty_xy);
    if (ctx->d_dirty_val) hipFree(ctx->d_dirty_val);
    ctx->d_dirty_xy = nullptr;
    ctx->d_dirty_val = nullptr;
    ctx->dirty_cap = 0;
    int cap = 1024;
    while (cap < n) cap *= 2;
    SLAMHIP_CHECK(hipMalloc(&ctx->d_dirty_xy, sizeof(int) * 2 * cap));
    SLAMHIP_CHECK(hipMalloc(&ctx->d_dirty_val, sizeof(double) * 4 * cap));
    ctx->dirty_cap = cap;
  }
  SLAMHIP_CHECK(hipMemcpyAsync(ctx->d_dirty_xy, coords_xy, sizeof(int) * 2 * n, hipMemcpyHostToDevice,
                               ctx->stream));
  SLAMHIP_CHECK(hipMemcpyAsync(ctx->d_dirty_val, payloads, sizeof(double) * sh * n,
                               hipMemcpyHostToDevice, ctx->stream));
  SLAMHIP_CHECK(launch_scatter_cells(m->d_payload, m->pitch, cd, sh, n, ctx->d_dirty_xy,
                                     ctx->d_dirty_val, ctx->stream));
  if (m->nbr_ok)
    SLAMHIP_CHECK(launch_nbr_cells(m->d_payload, m->width, m->height, m->pitch, m->nbr_th, n, ctx->d_dirty_xy, ctx->stream));
  if (m->prob_ok)
    SLAMHIP_CHECK(launch_prob_cells(m->d_payload, m->d_prob, m->width, m->height, m->pitch, n, ctx->d_dirty_xy, ctx->stream));
  SLAMHIP_CHECK(hipStreamSynchronize(ctx->stream));
  return SLAMHIP_OK;
}

// ---------------------------------------------------------------------------------- scan
// The scan's staging: two pinned buffers taking turns, packed {range, cos, sin, weight, factor} x `stride` doubles.
// acquire: room for n_max points, waits until the pull of two uploads ago has read the buffer (a word in pinned
// memory the pull kernel writes: no event calls), returns the buffer and its stride.
// commit: n <= n_max points are in the buffer -- host copies of weights / factors, the total weight, and the pull
// kernel queued on the context's stream.
static int scan_stage_acquire(slamhip_ctx *ctx, int n_max, double **st, size_t *stride) {
  SLAMHIP_CHECK(hipSetDevice(ctx->device));
  if (n_max > ctx->scan_cap) {
    SLAMHIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->stream_b) SLAMHIP_CHECK(hipStreamSynchronize(ctx->stream_b));
    if (ctx->d_scan) hipFree(ctx->d_scan);
    ctx->d_scan = nullptr;
    ctx->scan_cap = 0;
    int cap = 2048;
    while (cap < n_max) cap *= 2;
    SLAMHIP_CHECK(hipMalloc(&ctx->d_scan, sizeof(double) * 5 * cap));
    for (int k = 0; k < 2; ++k) {
      if (ctx->h_scan_stage[k]) hipHostFree(ctx->h_scan_stage[k]);
      ctx->h_scan_stage[k] = nullptr;
      SLAMHIP_CHECK(hipHostMalloc(&ctx->h_scan_stage[k], sizeof(double) * 5 * cap, hipHostMallocMapped));
      if (!ctx->scan_stage_done[k]) SLAMHIP_CHECK(hipEventCreateWithFlags(&ctx->scan_stage_done[k], hipEventDisableTiming));
      ctx->scan_pull_seq[k] = 0;
    }
    if (!ctx->h_scan_pulled) {
      SLAMHIP_CHECK(hipHostMalloc(&ctx->h_scan_pulled, sizeof(unsigned) * 2, hipHostMallocMapped | hipHostMallocCoherent));
      ctx->h_scan_pulled[0] = ctx->h_scan_pulled[1] = 0;
      SLAMHIP_CHECK(hipMalloc(&ctx->d_scan_pull_count, sizeof(unsigned)));
      // (on the context's stream: hipMemset on the null stream is not ordered with it -- the first pull's workgroups
      // were seen counting into a word that was cleared under them)
      SLAMHIP_CHECK(hipMemsetAsync(ctx->d_scan_pull_count, 0, sizeof(unsigned), ctx->stream));
    }
    ctx->scan_cap = cap;
  }
  const int turn = ctx->scan_stage_turn;
  if (ctx->scan_pull_seq[turn] != 0) {
    volatile unsigned *pulled = ctx->h_scan_pulled + turn;
    unsigned long long spins = 0;
    while (*pulled != ctx->scan_pull_seq[turn]) {
      __builtin_ia32_pause();
      if ((++spins & 0xfffffull) == 0) {
        hipError_t qe = hipStreamQuery(ctx->stream);
        if (qe != hipSuccess && qe != hipErrorNotReady) return slamhip::hip_fail(qe, "scan pull kernel");
        if (qe == hipSuccess && *pulled != ctx->scan_pull_seq[turn]) {
          char msg[200];
          std::snprintf(msg, sizeof msg, "internal: a scan pull never reported back (buffer %d: pull %u queued, %u / %u reported, %u pulls so far)",
                        turn, ctx->scan_pull_seq[turn], ctx->h_scan_pulled[0], ctx->h_scan_pulled[1], ctx->scan_pull_next);
          return invalid(msg);
        }
      }
    }
  }
  *st = ctx->h_scan_stage[turn];
  *stride = (size_t)((n_max + 7) & ~7);
  return SLAMHIP_OK;
}

static int scan_stage_commit(slamhip_ctx *ctx, int n, double *st, size_t stride) {
  const double *weight = st + 3 * stride, *factor = st + 4 * stride;
  ctx->h_weight.assign(weight, weight + n);
  ctx->h_factor.assign(factor, factor + n);
  // total_weight accumulates in beam order and does not depend on the pose
  // (weighted_mean_point_probability_spe.h:125)
  double tot_w = 0;
  for (int i = 0; i < n; ++i) tot_w += weight[i];
  ctx->scan_tot_w = tot_w;
  ctx->scan_n = n;
  ctx->scan_ptr = ctx->d_scan;
  ctx->scan_stride = stride;
  ctx->h_scan_angle.clear();  // (the angles of the scan before: slamhip_scan_set_angles)
  ctx->scan_angle_on_device = false;
  const int turn = ctx->scan_stage_turn;
  ctx->scan_stage_turn ^= 1;
  unsigned seq = ++ctx->scan_pull_next;
  if (seq == 0) seq = ++ctx->scan_pull_next;
  ctx->scan_pull_seq[turn] = seq;
  // (one pull over the five stretches, gaps included: 5 x stride doubles are 43 KB at 1080 beams)
  SLAMHIP_CHECK(launch_scan_pull(st, ctx->d_scan, 4 * stride + (size_t)n, ctx->d_scan_pull_count,
                                 ctx->h_scan_pulled + turn, seq, ctx->stream));
  if (ctx->stream_b) {  // the second launch lane waits for the scan
    SLAMHIP_CHECK(hipEventRecord(ctx->scan_stage_done[turn], ctx->stream));
    SLAMHIP_CHECK(hipStreamWaitEvent(ctx->stream_b, ctx->scan_stage_done[turn], 0));
  }
  return SLAMHIP_OK;
}

int slamhip_scan_upload(slamhip_ctx *ctx, int n, const double *range, const double *cos_a,
                        const double *sin_a, const double *weight, const double *factor) {
  if (!ctx) return invalid("null ctx");
  if (n <= 0 || !range || !cos_a || !sin_a || !weight) return invalid("bad scan");
  double *st = nullptr;
  size_t c = 0;
  int rc = scan_stage_acquire(ctx, n, &st, &c);
  if (rc) return rc;
  const size_t bytes = sizeof(double) * n;
  std::memcpy(st, range, bytes);
  std::memcpy(st + c, cos_a, bytes);
  std::memcpy(st + 2 * c, sin_a, bytes);
  std::memcpy(st + 3 * c, weight, bytes);
  if (factor) std::memcpy(st + 4 * c, factor, bytes);
  else for (int i = 0; i < n; ++i) st[4 * c + i] = 1.0;
  return scan_stage_commit(ctx, n, st, c);
}

int slamhip_scan_store(slamhip_ctx *ctx, int slot, int n, const double *range, const double *cos_a,
                       const double *sin_a, const double *weight, const double *factor) {
  if (!ctx) return invalid("null ctx");
  if (slot < 0 || slot >= 4096) return invalid("scan slot out of range (0..4095)");
  if (n <= 0 || !range || !cos_a || !sin_a || !weight) return invalid("bad scan");
  SLAMHIP_CHECK(hipSetDevice(ctx->device));
  if ((int)ctx->scan_slots.size() <= slot) ctx->scan_slots.resize(slot + 1);
  slamhip_ctx::ScanSlot &sl = ctx->scan_slots[slot];
  // work queued on the context may still read the slot (or the scan it is selected as)
  SLAMHIP_CHECK(hipStreamSynchronize(ctx->stream));
  if (ctx->stream_b) SLAMHIP_CHECK(hipStreamSynchronize(ctx->stream_b));
  if (n > sl.cap) {
    const bool selected = ctx->scan_ptr == sl.d && sl.d;
    if (sl.d) hipFree(sl.d);
    sl.d = nullptr;
    sl.cap = 0;
    const int cap = (n + 7) & ~7;
    SLAMHIP_CHECK(hipMalloc(&sl.d, sizeof(double) * 5 * cap));
    sl.cap = cap;
    if (selected) ctx->scan_ptr = sl.d;
  }
  sl.n = n;
  sl.w.assign(weight, weight + n);
  if (factor) sl.f.assign(factor, factor + n);
  else sl.f.assign(n, 1.0);
  double tot_w = 0;  // in beam order (weighted_mean_point_probability_spe.h:125)
  for (int i = 0; i < n; ++i) tot_w += weight[i];
  sl.tot_w = tot_w;
  const size_t bytes = sizeof(double) * n, c = (size_t)sl.cap;
  SLAMHIP_CHECK(hipMemcpy(sl.d, range, bytes, hipMemcpyHostToDevice));
  SLAMHIP_CHECK(hipMemcpy(sl.d + c, cos_a, bytes, hipMemcpyHostToDevice));
  SLAMHIP_CHECK(hipMemcpy(sl.d + 2 * c, sin_a, bytes, hipMemcpyHostToDevice));
  SLAMHIP_CHECK(hipMemcpy(sl.d + 3 * c, weight, bytes, hipMemcpyHostToDevice));
  SLAMHIP_CHECK(hipMemcpy(sl.d + 4 * c, sl.f.data(), bytes, hipMemcpyHostToDevice));
  if (ctx->scan_ptr == sl.d) {  // the selected scan was rewritten in place
    ctx->scan_n = n;
    ctx->scan_tot_w = tot_w;
    ctx->scan_stride = c;
    ctx->h_weight = sl.w;
    ctx->h_factor = sl.f;
  }
  return SLAMHIP_OK;
}

int slamhip_scan_select(slamhip_ctx *ctx, int slot) {
  if (!ctx) return invalid("null ctx");
  if (slot < 0 || slot >= (int)ctx->scan_slots.size() || !ctx->scan_slots[slot].d || ctx->scan_slots[slot].n <= 0)
    return invalid("no scan stored in that slot");
  const slamhip_ctx::ScanSlot &sl = ctx->scan_slots[slot];
  ctx->scan_ptr = sl.d;
  ctx->scan_stride = (size_t)sl.cap;
  ctx->scan_n = sl.n;
  ctx->h_scan_angle.clear();
  ctx->scan_angle_on_device = false;
  ctx->scan_tot_w = sl.tot_w;
  ctx->h_weight.assign(sl.w.begin(), sl.w.end());
  ctx->h_factor.assign(sl.f.begin(), sl.f.end());
  return SLAMHIP_OK;
}

int slamhip_beam_trig_raw(int n, const double *angle, double *cos_out, double *sin_out) {
  if (n < 0 || !angle || !cos_out || !sin_out) return invalid("bad arguments");
  for (int i = 0; i < n; ++i) ::sincos(angle[i], &sin_out[i], &cos_out[i]);
  return SLAMHIP_OK;
}

int slamhip_beam_trig_cached(int n, const double *angle, double a_min, double a_max, double a_inc,
                             double *cos_out, double *sin_out) {
  if (n < 0 || !angle || !cos_out || !sin_out || !(a_inc > 0)) return invalid("bad arguments");
  // the table of the last (a_min, a_max, a_inc) is kept: a world loop asks for the same one twice per scan, and
  // building it is a libm sincos per entry
  struct Table {
    double a_min = 0, a_max = 0, a_inc = 0;
    std::vector<double> ts, tc;
  };
  static thread_local Table tab;
  if (tab.ts.empty() || tab.a_min != a_min || tab.a_max != a_max || tab.a_inc != a_inc) {
    tab.ts.clear();
    tab.tc.clear();
    for (double a = a_min; a < a_max; a += a_inc) {  // accumulating loop, as the provider builds it
      double sv, cv;
      ::sincos(a, &sv, &cv);  // fused pair, as in the reference build (see score_staged)
      tab.ts.push_back(sv);
      tab.tc.push_back(cv);
    }
    tab.a_min = a_min;
    tab.a_max = a_max;
    tab.a_inc = a_inc;
  }
  const std::vector<double> &ts = tab.ts, &tc = tab.tc;
  for (int i = 0; i < n; ++i) {
    const int idx = (int)std::round((angle[i] - a_min) / a_inc);
    if (idx < 0 || idx >= (int)ts.size()) return invalid("scan angle outside the trig table");
    cos_out[i] = tc[idx];
    sin_out[i] = ts[idx];
  }
  return SLAMHIP_OK;
}

int slamhip_filter_scan(int n, const double *range, const double *angle, const int *is_occ,
                        int trig_mode, double a_min, double a_delta, int table_n,
                        const double *tab_sin, const double *tab_cos, const double pose[3],
                        unsigned skip_rate, double max_range, int bounded, int width, int height,
                        int origin_x, int origin_y, double scale, int *kept_idx, int *kept_n) {
  if (n < 0 || !range || !angle || !pose || !kept_idx || !kept_n) return invalid("bad arguments");
  if (trig_mode == SLAMHIP_TRIG_CACHED && (!tab_sin || !tab_cos || table_n <= 0))
    return invalid("cached trig mode needs the table");
  constexpr double eps = std::numeric_limits<double>::epsilon();
  double sb, cb;
  ::sincos(pose[2], &sb, &cb);
  int kept = 0;
  for (int i = 0; i < n; ++i) {
    if (skip_rate && (unsigned(i) % skip_rate)) continue;  // keeps every k-th point (Q8)
    double c, s;
    if (trig_mode == SLAMHIP_TRIG_CACHED) {
      const int idx = (int)std::round((angle[i] - a_min) / a_delta);
      if (idx < 0 || idx >= table_n) return invalid("scan angle outside the trig table");
      c = cb * tab_cos[idx] - sb * tab_sin[idx];
      s = sb * tab_cos[idx] + cb * tab_sin[idx];
    } else {
      ::sincos(pose[2] + angle[i], &s, &c);
    }
    const double wx = pose[0] + range[i] * c, wy = pose[1] + range[i] * s;
    const int cx = (int)std::floor(wx / scale), cy = (int)std::floor(wy / scale);
    bool has_cell = true;
    if (bounded) {
      const int ix = cx + origin_x, iy = cy + origin_y;
      has_cell = 0 <= ix && ix < width && 0 <= iy && iy < height;
    }
    const bool too_far = (0.0 < max_range + eps) && (max_range < range[i] + eps);
    if ((is_occ && !is_occ[i]) || !has_cell || too_far) continue;
    kept_idx[kept++] = i;
  }
  *kept_n = kept;
  return SLAMHIP_OK;
}

int slamhip_scan_weights(int kind, int n, const double *range, const double *angle, double *out) {
  if (n < 0 || !range || !angle || !out) return invalid("bad arguments");
  if (kind == 0) {
    const double w = 1.0 / n;
    for (int i = 0; i < n; ++i) out[i] = w;
    return SLAMHIP_OK;
  }
  if (kind == 1) {
    for (int i = 0; i < n; ++i) {
      double sv, cv;
      ::sincos(angle[i], &sv, &cv);
      const double ac = std::abs(cv);
      double w = std::abs(sv) + ac;
      if (0.9 < ac) w = 3;
      else if (0.8 < ac) w = 2;
      out[i] = w * std::sqrt(range[i]);
    }
    return SLAMHIP_OK;
  }
  if (kind == 2) {
    constexpr int NB = 20;
    unsigned hist[NB] = {0};
    std::vector<double> dirs(n > 0 ? n : 1, 0.0);
    const double step = (180 * M_PI / 180) / NB;
    for (int i = 1; i < n; ++i) {
      double s1, c1, s0, c0;
      ::sincos(angle[i], &s1, &c1);
      ::sincos(angle[i - 1], &s0, &c0);
      const double d_x = range[i] * c1 - range[i - 1] * c0;
      const double d_y = range[i] * s1 - range[i - 1] * s0;
      double a = 0;
      if (d_y != 0) {
        a = std::acos(d_x / std::sqrt(d_x * d_x + d_y * d_y));
        if (d_y < 0 && d_x != 0) a = M_PI - a;
      }
      const size_t bin = (size_t)std::floor(a / step);
      if (bin >= NB) return invalid("angle histogram bin out of range (reference asserts here)");
      hist[bin]++;
      dirs[i] = a;
    }
    for (int i = 0; i < n; ++i) {
      const unsigned v = i == 0 ? (unsigned)n : hist[(size_t)std::floor(dirs[i] / step)];
      out[i] = 1.0 / v;
    }
    return SLAMHIP_OK;
  }
  return invalid("unknown weighting kind");
}

// The per-scan host half of a match in one call, from the RAW scan: WeightedMeanPointProbabilitySPE::filter_scan
// (weighted_mean_point_probability_spe.h:75-95,136-141) at the initial pose, the scan-point weights of the filtered
// scan (:21-60), the beams' cos / sin as the scan's trig provider tabulates them, and the upload.  Through the adapter
// classes this half was the larger one of a resident world's scan (70 of 140 us): the reference's filter_scan calls
// libm twice per point for an end point whose only use is has_cell() -- always true on an unbounded map -- and the
// beams' sincos were made again for every scan although a scanner's angles never change.  Here nothing is computed
// that the decision does not need: no end point unless the map is bounded, angle-only quantities cached until the
// angle array changes (compared by content).
int slamhip_scan_filter_upload(slamhip_ctx *ctx, int map_id, int n, const double *range, const double *angle,
                               const int *is_occ, const double *factor, int trig_mode, double a_min, double a_max,
                               double a_inc, const double pose[3], unsigned skip_rate, double max_range, int bounded,
                               int weighting, int *kept_n, int *kept_idx) {
  if (!ctx || n <= 0 || !range || !angle || !pose) return invalid("bad arguments");
  if (weighting < 0 || weighting > 2) return invalid("unknown weighting kind");
  DeviceMap *m = get_map(ctx, map_id);
  if (bounded && !m) return invalid("unknown map id");
  slamhip_ctx::ScanPrep &sp = ctx->scan_prep;
  const bool same = (int)sp.angle.size() == n && sp.trig_mode == trig_mode && sp.a_min == a_min && sp.a_max == a_max &&
                    sp.a_inc == a_inc && std::memcmp(sp.angle.data(), angle, sizeof(double) * n) == 0;
  if (!same) {
    sp.angle.assign(angle, angle + n);
    sp.trig_mode = trig_mode;
    sp.a_min = a_min;
    sp.a_max = a_max;
    sp.a_inc = a_inc;
    sp.cos_a.resize(n);
    sp.sin_a.resize(n);
    int rc = trig_mode == SLAMHIP_TRIG_CACHED
                 ? slamhip_beam_trig_cached(n, angle, a_min, a_max, a_inc, sp.cos_a.data(), sp.sin_a.data())
                 : slamhip_beam_trig_raw(n, angle, sp.cos_a.data(), sp.sin_a.data());
    if (rc) {
      sp.angle.clear();
      return rc;
    }
    // VinySlamSPW's angular factor (weighted_mean_point_probability_spe.h:47-60): the weight is this times sqrt(range)
    sp.viny_f.resize(n);
    for (int i = 0; i < n; ++i) {
      double sv, cv;
      ::sincos(angle[i], &sv, &cv);
      const double ac = std::abs(cv);
      double w = std::abs(sv) + ac;
      if (0.9 < ac) w = 3;
      else if (0.8 < ac) w = 2;
      sp.viny_f[i] = w;
    }
  }
  // ---- filter_scan: keeps point i iff (skip_rate == 0 or i % skip_rate == 0), occupied, its end point's cell in the
  // map, and not beyond the usable range (Q8 - Q10)
  constexpr double eps = std::numeric_limits<double>::epsilon();
  sp.kept.clear();
  double sb = 0, cb = 1;
  if (bounded) ::sincos(pose[2], &sb, &cb);
  for (int i = 0; i < n; ++i) {
    if (skip_rate && (unsigned(i) % skip_rate)) continue;
    if (is_occ && !is_occ[i]) continue;
    const bool too_far = (0.0 < max_range + eps) && (max_range < range[i] + eps);
    if (too_far) continue;
    if (bounded) {
      double c, s;
      if (trig_mode == SLAMHIP_TRIG_CACHED) {
        c = cb * sp.cos_a[i] - sb * sp.sin_a[i];
        s = sb * sp.cos_a[i] + cb * sp.sin_a[i];
      } else {
        ::sincos(pose[2] + angle[i], &s, &c);
      }
      const double wx = pose[0] + range[i] * c, wy = pose[1] + range[i] * s;
      const int ix = (int)std::floor(wx / m->scale) + m->origin_x, iy = (int)std::floor(wy / m->scale) + m->origin_y;
      if (!(0 <= ix && ix < m->width && 0 <= iy && iy < m->height)) continue;
    }
    sp.kept.push_back(i);
  }
  const int k = (int)sp.kept.size();
  if (kept_n) *kept_n = k;
  if (kept_idx) std::memcpy(kept_idx, sp.kept.data(), sizeof(int) * k);
  if (k == 0) {
    ctx->scan_n = 0;  // (scoring without a scan fails; the reference scores NaN: the adapter handles an empty scan itself)
    return SLAMHIP_OK;
  }
  // straight into the pinned staging buffer the pull kernel reads (no intermediate arrays)
  double *st = nullptr;
  size_t c = 0;
  int rc = scan_stage_acquire(ctx, k, &st, &c);
  if (rc) return rc;
  double *r_ = st, *c_ = st + c, *s_ = st + 2 * c, *w_ = st + 3 * c, *f_ = st + 4 * c;
  for (int q = 0; q < k; ++q) {
    const int i = sp.kept[q];
    r_[q] = range[i];
    c_[q] = sp.cos_a[i];
    s_[q] = sp.sin_a[i];
    f_[q] = factor ? factor[i] : 1.0;
  }
  if (weighting == 0) {
    const double w = 1.0 / k;  // EvenSPW (:21-32)
    for (int q = 0; q < k; ++q) w_[q] = w;
  } else if (weighting == 1) {
    for (int q = 0; q < k; ++q) w_[q] = sp.viny_f[sp.kept[q]] * std::sqrt(r_[q]);
  } else {
    sp.a.resize(k);
    sp.r.assign(r_, r_ + k);
    sp.w.resize(k);
    for (int q = 0; q < k; ++q) sp.a[q] = angle[sp.kept[q]];
    rc = slamhip_scan_weights(2, k, sp.r.data(), sp.a.data(), sp.w.data());
    if (rc) return rc;
    std::memcpy(w_, sp.w.data(), sizeof(double) * k);
  }
  rc = scan_stage_commit(ctx, k, st, c);
  if (rc) return rc;
  // the kept points' angles, for SLAMHIP_POSE_TRIG_RAW_EXACT (host side only: they go to HBM when an exact call asks)
  ctx->h_scan_angle.resize(k);
  for (int q = 0; q < k; ++q) ctx->h_scan_angle[q] = angle[sp.kept[q]];
  return SLAMHIP_OK;
}

int slamhip_scan_set_angles(slamhip_ctx *ctx, int n, const double *angle) {
  if (!ctx || !angle) return invalid("bad arguments");
  if (n != ctx->scan_n || n <= 0) return invalid("slamhip_scan_set_angles: n is not the current scan's point count");
  ctx->h_scan_angle.assign(angle, angle + n);
  ctx->scan_angle_on_device = false;
  return SLAMHIP_OK;
}

int slamhip_libm_variant(int *variant) {
  if (!variant) return invalid("null variant");
  *variant = libm_variant();
  return SLAMHIP_OK;
}

int slamhip_libm_eval(slamhip_ctx *ctx, int variant, int fn, int n, const double *x, double *out) {
  if (!ctx || n < 0 || (n > 0 && (!x || !out)) || fn < 0 || fn > 2 || (variant != 0 && variant != 1))
    return invalid("bad arguments");
  if (n == 0) return SLAMHIP_OK;
  SLAMHIP_CHECK(hipSetDevice(ctx->device));
  double *d = nullptr;
  SLAMHIP_CHECK(hipMalloc(&d, sizeof(double) * 2 * (size_t)n));
  hipError_t e = hipMemcpyAsync(d, x, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = launch_libm_eval(variant == 1, fn, d, d + n, n, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(out, d + n, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  hipFree(d);
  if (e != hipSuccess) return hip_fail(e, "slamhip_libm_eval");
  return SLAMHIP_OK;
}

// ObservationMappingQualityEstimator::quality (grid_map_scan_adders.h:17-43): IdleOMQE, or
// AngleHistogramResiprocalOMQE = 1 / AngleHistogram::value -- the arithmetic of the `ahr` scan-point weighting above
// (weighted_mean_point_probability_spe.h:34-45 is the same reciprocal of the same histogram)
int slamhip_omqe_quality(int kind, int n, const double *range, const double *angle, double *out) {
  if (n < 0 || !range || !angle || !out) return invalid("bad arguments");
  if (kind == 0) {
    for (int i = 0; i < n; ++i) out[i] = 1.0;
    return SLAMHIP_OK;
  }
  if (kind == 1) return slamhip_scan_weights(2, n, range, angle, out);
  return invalid("unknown observation quality estimator");
}

// ---------------------------------------------------------------------------------- scoring
int slamhip_score_poses(slamhip_ctx *ctx, int map_id, const slamhip_spe_cfg *cfg, int n_poses,
                        const double *poses_xyt, double *scores_out) {
  if (!ctx) return invalid("null ctx");
  if (n_poses < 0 || (n_poses > 0 && (!poses_xyt || !scores_out))) return invalid("bad pose batch");
  if (n_poses == 0) return SLAMHIP_OK;
  SLAMHIP_CHECK(hipSetDevice(ctx->device));
  int rc = ensure_pose_capacity(ctx, n_poses);
  if (rc) return rc;
  std::memcpy(ctx->h_poses, poses_xyt, sizeof(double) * 3 * n_poses);
  rc = score_staged(ctx, map_id, cfg, n_poses);
  if (rc) return rc;
  if (cfg->oope == SLAMHIP_OOPE_GMAPPING) gm_carry_fixup(ctx, n_poses);
  std::memcpy(scores_out, ctx->h_scores, sizeof(double) * n_poses);
  return SLAMHIP_OK;
}

int slamhip_score_poses_device(slamhip_ctx *ctx, int map_id, const slamhip_spe_cfg *cfg,
                               int n_poses, const double *d_poses_xyt, double *d_scores_out) {
  DeviceMap *m = get_map(ctx, map_id);
  if (!m) return invalid("unknown map id");
  int rc = check_cfg(*m, cfg);
  if (rc) return rc;
  if (n_poses <= 0) return SLAMHIP_OK;
  if (!d_poses_xyt || !d_scores_out) return invalid("null device buffers");
  if (cfg->pose_trig != SLAMHIP_POSE_TRIG_DEVICE)
    return invalid("device-resident poses use device sincos (pose_trig = DEVICE): the host-trig and exact modes take host poses");
  if (cfg->oope == SLAMHIP_OOPE_GMAPPING && cfg->sum_order == SLAMHIP_SUM_SEQUENTIAL)
    return invalid("the GMapping OOPE's beam-order mode keeps the reference's cache in call order: slamhip_score_poses");
  SLAMHIP_CHECK(hipSetDevice(ctx->device));
  ScoreArgs a;
  rc = fill_args(ctx, *m, cfg, n_poses, d_poses_xyt, nullptr, d_scores_out, &a);
  if (rc) return rc;
  if (cfg->oope == SLAMHIP_OOPE_GMAPPING) {
    rc = ensure_pose_capacity(ctx, n_poses);
    if (rc) return rc;
    a.gm_info = nullptr;  // in-pose runs only; cross-pose carry needs the host path
  }
  return launch_timed(ctx, a, *m, cfg, ctx->stream);
}

int slamhip_gm_cache_reset(slamhip_ctx *ctx) {
  if (!ctx) return invalid("null ctx");
  ctx->gm_cx = ctx->gm_cy = 0;
  ctx->gm_prob = -1.0;
  return SLAMHIP_OK;
}

int slamhip_gm_cache_get(slamhip_ctx *ctx, int *cell_xy, double *prob) {
  if (!ctx || !cell_xy || !prob) return invalid("bad arguments");
  cell_xy[0] = ctx->gm_cx;
  cell_xy[1] = ctx->gm_cy;
  *prob = ctx->gm_prob;
  return SLAMHIP_OK;
}

int slamhip_gm_cache_set(slamhip_ctx *ctx, const int *cell_xy, double prob) {
  if (!ctx || !cell_xy) return invalid("bad arguments");
  ctx->gm_cx = cell_xy[0];
  ctx->gm_cy = cell_xy[1];
  ctx->gm_prob = prob;
  return SLAMHIP_OK;
}

int slamhip_profile_enable(slamhip_ctx *ctx, int on) {
  if (!ctx) return invalid("null ctx");
  ctx->profile = on != 0;
  return SLAMHIP_OK;
}

int slamhip_profile_read(slamhip_ctx *ctx, double *kernel_ms_total, long long *launches,
                         long long *units, int reset) {
  if (!ctx) return invalid("null ctx");
  int rc = profile_resolve(ctx);
  if (rc) return rc;
  if (kernel_ms_total) *kernel_ms_total = ctx->prof_ms;
  if (launches) *launches = ctx->prof_launches;
  if (units) *units = ctx->prof_units;
  if (reset) {
    ctx->prof_ms = 0;
    ctx->prof_launches = 0;
    ctx->prof_units = 0;
  }
  return SLAMHIP_OK;
}

int slamhip_profile_read_map_update(slamhip_ctx *ctx, double *ms_total, long long *calls, long long *records,
                                    int reset) {
  if (!ctx) return invalid("null ctx");
  int rc = profile_resolve(ctx);
  if (rc) return rc;
  if (ms_total) *ms_total = ctx->prof_k6_ms;
  if (calls) *calls = ctx->prof_k6_calls;
  if (records) *records = ctx->prof_k6_records;
  if (reset) {
    ctx->prof_k6_ms = 0;
    ctx->prof_k6_calls = ctx->prof_k6_records = 0;
  }
  return SLAMHIP_OK;
}

// ---------------------------------------------------------------------------------- particle filter
int slamhip_pf_normalize(int n, double *w) {
  if (n < 0 || (n > 0 && !w)) return invalid("bad arguments");
  double total = 0;
  for (int i = 0; i < n; ++i) total += w[i];
  for (int i = 0; i < n; ++i) w[i] = w[i] / total;
  return SLAMHIP_OK;
}

int slamhip_pf_resampling_is_required(int n, const double *w, int *required) {
  if (n < 0 || (n > 0 && !w) || !required) return invalid("bad arguments");
  double sq_sum = 0;
  for (int i = 0; i < n; ++i) sq_sum += w[i] * w[i];
  const double effective = 1.0 / sq_sum;
  *required = effective * 2 < (double)(size_t)n;
  return SLAMHIP_OK;
}

int slamhip_pf_resample(int n, const double *w, uint32_t seed, unsigned *out_idx) {
  if (n < 0 || (n > 0 && (!w || !out_idx))) return invalid("bad arguments");
  // same libstdc++ machinery as the reference: mt19937 + uniform_real_distribution<>(0, 1),
  // linear CDF scan, index stays 0 when the draw is not below the total (Q24)
  std::mt19937 engine(seed);
  std::uniform_real_distribution<> uniform(0, 1);
  for (int i = 0; i < n; ++i) {
    out_idx[i] = 0;
    const double sample = uniform(engine);
    double total_w = 0;
    for (int j = 0; j < n; ++j) {
      total_w += w[j];
      if (sample < total_w) {
        out_idx[i] = (unsigned)j;
        break;
      }
    }
  }
  return SLAMHIP_OK;
}

int slamhip_pf_heaviest(int n, const double *w, int *index) {
  if (n <= 0 || !w || !index) return invalid("bad arguments");
  int h = -1;
  for (int i = 0; i < n; ++i) {
    if (h >= 0 && w[i] < w[h]) continue;  // last of equal maxima wins
    h = i;
  }
  *index = h;
  return SLAMHIP_OK;
}

}  // extern "C"
