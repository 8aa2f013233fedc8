// Internal declarations shared by the HIP kernels (score_kernels.hip) and the host side
// of the C-ABI (slamhip_api.cpp, matchers.cpp).  Not installed; the public ABI is include/slamhip.h.
#pragma once

// Testing and debugging hooks (in-kernel wall-clock stamps, a workgroup that leaves at once, injected failures, a stalled
// stream, a pretend-small trace buffer) exist only in libslamhip_testing.so, which csrc/Makefile builds from the same
// sources with -DSLAMHIP_TESTING and which the tests that need a hook load; the shipped libslamhip.so exports none of
// them and its kernels carry no branch for them.
#ifdef SLAMHIP_TESTING
#define SLAMHIP_STAMPS_ON(expr) (expr)
#else
#define SLAMHIP_STAMPS_ON(expr) false
#endif


#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "slamhip.h"

namespace slamhip {

// ---- views passed to kernels by value -----------------------------------------------------
// Map window in HBM: row-major [height][pitch] cells, each cell CELL_DOUBLES(model) doubles
// (OCC 1, TBM 4, GMAPPING 4 = prob_occ, obst.x, obst.y, pad -> 32-byte aligned gathers).
// The pad of a GMAPPING cell of a DENSE window holds, in its low dword, the cell's NEIGHBOURHOOD MASK once a GMapping
// scorer has asked for it (nbr_ok): bit 3 (dx + 1) + (dy + 1) = "the cell at (+dx, +dy) is inside the window and full",
// full = !(prob_occ < fullness_th) -- the scorer's own test.  One 4-byte load then tells a beam which of the nine
// cells of its 3 x 3 window it has to look at (gm_score_device.h); writers keep the masks (mu_cell_store flips the
// nine neighbours' bits when a cell changes sides; uploads re-derive them) or drop them (DeviceMap::nbr_ok).
struct MapView {
  const double *payload;
  int width, height, pitch;
  int origin_x, origin_y;
  double scale, inv_scale;  // inv_scale = RN(1 / scale), see to_cell()
  double unknown[4];
  int nbr_ok;  // the neighbourhood masks of this (dense, GMAPPING) window are valid for the scorer's threshold
  int reserved_;
};

// TbmBaseCell::discrepancy of the scorer's fixed observation, 1 - it: the per-beam probability of a TBM cell under the
// discrepancy OIE (tbm_grid_cells.h:21-35, transferable_belief_model.h:102-143; SURVEY Q18: the observation is always
// (u, e, o, c) = (0, 0, 1, 0), so the value is a PURE FUNCTION OF THE CELL).  One definition for the scoring kernels
// (cell_probability<TBM>), the probability plane's writers and the host (the prototype cell's value): the same
// operations in the same order, hence the same bits wherever it is evaluated (-ffp-contract=off).
__host__ __device__ inline double tbm_discrepancy_probability(double U, double E, double O, double Cc) {
  // that = aoo2tbm(obstacle AOO) = (u,e,o,c) = (0,0,1,0); conjunctive(that, cell) before
  // normalisation = (0, 0, U+O, E+C); normalize() divides by the total mass.
  const double d_occ = __builtin_fabs(1.0 - O);
  const double t2 = U + O, t3 = E + Cc;
  const double tot = t2 + t3;
  const double conflict = (tot == 0.0) ? 0.0 : t3 / tot;
  const double unknown = U / 2.0;
  const double known = 1 - unknown;
  const double known_discrepancy = known * (conflict + d_occ) / 2.0;
  return 1.0 - (unknown / 2 + known_discrepancy);
}

// Filtered scan, structure-of-arrays (coalesced per-beam loads), tot_w = sequential sum of
// weights in beam order computed once on the host (pose independent).
struct ScanView {
  const double *range, *cos_a, *sin_a, *weight, *factor;
  int n;
  double tot_w;
};

struct GmParams {
  double fullness_th;
  int window;
};

// per-pose side outputs of the GMapping kernel for the host carry-in fix-up
struct GmPoseInfo {
  int first_cx, first_cy;  // cell of beam 0
  int last_cx, last_cy;    // cell of the last beam
  double v0;               // fresh value of the first run
  double last_v;           // resolved value of the last run (before any carry-in)
  int run0_len;            // beams in the first run (they all hold v0)
  int last_head;           // beam index heading the last run (0 => the last run is run 0)
};

struct ScoreArgs {
  MapView map;
  ScanView scan;
  const double *poses;  // n_poses x 3 (x, y, theta)
  const double *pose_sc;  // optional n_poses x 2 (sin, cos) computed by the host; null = device
  double *scores;
  int n_poses;
  int poses_per_block;
  int oie;
  double area[4];
  GmParams gm;
  GmPoseInfo *gm_info;  // GMAPPING only
  double *terms;        // SEQUENTIAL order only: n_poses x scan.n scratch
  unsigned long long *fprints;  // K1, canonical sum only: per-pose term-vector fingerprints, 64 bits (null = none)
  // per-particle copy-on-write maps (K3 only, tile_pool.h): tile tables of all slots, the slot of
  // every pose; then map.payload = tile pool, map.pitch = tiles per table row, map.width/height =
  // the virtual extent in cells.  null = one dense window for all poses.
  const int *tables;
  const int *pose_slot;
  int table_stride;
  int xcd_blocks;  // set by launch_score: > 0 = the real block count of an XCD-chunked grid
  // fill_args: >= 0 = the model the kernels are instantiated for instead of the map's own -- a TBM map scored through
  // its probability plane looks like an OCC map under the occupancy OIE (DeviceMap::d_prob)
  int model_override;
};

// tiles of the copy-on-write maps: 128 x 128 cells like the reference's LazyTiledGridMap
// (src/core/maps/lazy_tiled_grid_map.h:24-26), row-major inside a tile
constexpr int kTileShift = 7;
constexpr int kTileSide = 1 << kTileShift;
constexpr int kTileMask = kTileSide - 1;
constexpr int kTileCells = kTileSide * kTileSide;
// the settle state of a pool cell (tile_pool.h TilePool::d_state): 1 = its mean is +0; 3 = a free observation leaves +0
// but has to WRITE it -- the never-observed value (a negative unknown[0], bit for bit), or -0; 0 = anything else
constexpr int kTileStateWords = kTileCells / 16;
__host__ __device__ inline unsigned mu_settle_class(double c0, long long unknown_bits, int fresh_ok) {
  long long bits;
  __builtin_memcpy(&bits, &c0, 8);
  return bits == 0ll ? 1u : ((c0 == 0.0 || (fresh_ok && bits == unknown_bits)) ? 3u : 0u);
}

inline int cell_doubles(int model) { return model == SLAMHIP_CELL_OCC ? 1 : 4; }
inline int cell_stride_host(int model) {
  return model == SLAMHIP_CELL_TBM ? 4 : (model == SLAMHIP_CELL_GMAPPING ? 3 : 1);
}

// ---- launchers (score_kernels.hip) ---------------------------------------------------------
// ev_start / ev_stop (optional, single-kernel orders only): attached to the scoring dispatch
hipError_t launch_score(const ScoreArgs &a, int cell_model, int oope, int sum_order,
                        hipStream_t stream, hipEvent_t ev_start = nullptr,
                        hipEvent_t ev_stop = nullptr);
hipError_t launch_publish(unsigned *flag, unsigned seq, hipStream_t stream);
// ---- the libm-exact modes (exact_kernels.hip; csrc/libm_exact.h) ----
// cos / sin(theta_p + a_b) with the host glibc's bits for n_poses poses x n beams ([p * stride + b]) + (0, 1) per pose
hipError_t launch_exact_beam_trig(bool fma, const double *poses, int n_poses, const double *d_angle, int n, size_t stride,
                                  double *d_cos, double *d_sin, double *d_identity_sc, hipStream_t stream);
// the GMapping OOPE restated the plain way (exp per full cell, cache beam after beam and pose after pose, beam-order sum)
hipError_t launch_score_gmapping_exact(bool fma, const ScoreArgs &a, const double *d_angle, int raw_trig, void *d_cache,
                                       hipStream_t stream);
hipError_t launch_libm_eval(bool fma, int fn, const double *d_x, double *d_out, int n, hipStream_t stream);
// which build of sin / cos / exp the host's libm runs: 1 = glibc's FMA build, 0 = the plain one, -1 = neither matches
int libm_variant();
hipError_t launch_stall(int ms, hipStream_t stream);  // testing
// n_doubles (rounded up to two) from pinned host memory to HBM by a kernel; *h_flag = seq once the source has been read
// bytes rounded up to 16: both blocks must be that long
hipError_t launch_block_pull(const void *h_src, void *d_dst, size_t bytes, hipStream_t stream);
hipError_t launch_scan_pull(const double *h_src, double *d_dst, size_t n_doubles, unsigned *counter, unsigned *h_flag,
                            unsigned seq, hipStream_t stream);
// (both uploads write the host's stride_host doubles of a cell and leave the rest of it alone -- a GMAPPING cell's pad)
hipError_t launch_scatter_cells(double *payload, int pitch, int cell_dbl, int stride_host, int n,
                                const int *d_coords, const double *d_vals, hipStream_t stream);
hipError_t launch_repack_window(double *dst, int dst_pitch, int cell_dbl, const double *src,
                                int stride_host, int x0, int y0, int w, int h, hipStream_t stream);
hipError_t launch_fill_cells(double *dst, size_t n_cells, int cell_dbl, const double *unknown4,
                             hipStream_t stream);

// ---- host state ------------------------------------------------------------------------------
struct DeviceMap {
  bool bound = false;
  int cell_model = 0;
  int width = 0, height = 0, pitch = 0;
  int origin_x = 0, origin_y = 0;
  double scale = 1.0;
  double unknown[4] = {0, 0, 0, 0};
  double *d_payload = nullptr;
  size_t bytes = 0;
  // update-only cell state, allocated by the first map update (map_update.hip):
  // MeanProbabilityCell::_n (1 double) or GmappingBaseCell::_hits/_tries (2 doubles) per cell
  double *d_aux = nullptr;
  int aux_stride = 0;
  // slamhip_map_set_auto_grow: an update that reaches beyond the window re-binds it first (the reference's
  // unbounded maps grow inside update(), plain_grid_map.h:133-173)
  bool auto_grow = false;
  long grown = 0;  // number of such re-binds
  // neighbourhood masks in the cells' pads (MapView): valid for threshold nbr_th; built by map_nbr_masks()
  bool nbr_ok = false;
  double nbr_th = 0.0;
  double nbr_other_th = 0.0;  // the last threshold a scorer asked for that was NOT the masks' (fill_args)
  // TBM maps: the PROBABILITY PLANE -- one double per cell = tbm_discrepancy_probability of the cell (r06, VERDICT r5
  // item 5).  The 1-cell scorers gather 8 bytes from it instead of the 32-byte cell and ~16 flops + a division per
  // (pose, beam); derived by the first scorer call that finds none (map_prob_plane), then kept by every writer: K6's
  // mu_cell_store (MuArgs::prob), uploads and the dirty log re-derive what they wrote; a re-bind drops it.
  double *d_prob = nullptr;
  bool prob_ok = false;
};
// (re)derives the plane over [x0, x0 + w) x [y0, y0 + h) (clipped) / over n listed cells (d_coords: x, y pairs)
hipError_t launch_prob_build(const double *payload, double *prob, int width, int height, int pitch, int x0, int y0, int w, int h,
                             hipStream_t stream);
hipError_t launch_prob_cells(const double *payload, double *prob, int width, int height, int pitch, int n, const int *d_coords,
                             hipStream_t stream);
// cells whose stored value differs from the derived one (testing), added to *d_count
hipError_t launch_prob_check(const double *payload, const double *prob, int width, int height, int pitch,
                             unsigned long long *d_count, hipStream_t stream);
// (re)derives the masks of the cells of [x0, x0 + w) x [y0, y0 + h) (clipped to the window) from the occupancies
hipError_t launch_nbr_build(double *payload, int width, int height, int pitch, double th, int x0, int y0, int w, int h,
                            hipStream_t stream);
// ... and of the 3 x 3 neighbourhoods of n listed cells (d_coords: x, y pairs)
hipError_t launch_nbr_cells(double *payload, int width, int height, int pitch, double th, int n, const int *d_coords,
                            hipStream_t stream);
// cells whose stored mask differs from the one their neighbours' occupancies give (testing), added to *d_count
hipError_t launch_nbr_check(const double *payload, int width, int height, int pitch, double th, unsigned long long *d_count,
                            hipStream_t stream);

void mu_allow_scan_reuse(slamhip_ctx *ctx, bool on);  // map_update.hip
bool mu_set_deferred(slamhip_ctx *ctx, bool on);      // map_update.hip: queue plain updates without waiting (returns the old setting)
int mu_drain(slamhip_ctx *ctx, long long *n_updates, int *err);
void mu_release(slamhip_ctx *ctx);                    // map_update.hip: frees the context's K6 scratch
void shard_release(slamhip_ctx *ctx);                 // shard.cpp: leaves the RCCL group, frees its staging
void set_error(const std::string &msg);
int hip_fail(hipError_t e, const char *what);

}  // namespace slamhip

#define SLAMHIP_CHECK(expr)                                        \
  do {                                                             \
    hipError_t _e = (expr);                                        \
    if (_e != hipSuccess) return slamhip::hip_fail(_e, #expr);     \
  } while (0)

struct slamhip_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::vector<slamhip::DeviceMap> maps;
  // scan
  double *d_scan = nullptr;  // 5 arrays of scan_cap doubles
  // slamhip_scan_upload: the five arrays packed in pinned memory (two buffers taking turns, an event each) and
  // sent with ONE asynchronous copy -- no wait for the stream, which may still be busy with a queued map update
  double *h_scan_stage[2] = {nullptr, nullptr};
  hipEvent_t scan_stage_done[2] = {nullptr, nullptr};  // (recorded for the second launch lane only)
  int scan_stage_turn = 0;
  // ... pulled into HBM by a kernel (k_scan_pull): the sequence number of the pull that last read each buffer, what
  // the kernel reported back (pinned), its arrival counter
  unsigned scan_pull_seq[2] = {0, 0}, scan_pull_next = 0;
  unsigned *h_scan_pulled = nullptr;
  unsigned *d_scan_pull_count = nullptr;
  int scan_cap = 0, scan_n = 0;
  double scan_tot_w = 0.0;
  // the scan the kernels read: d_scan after slamhip_scan_upload, a stored scan after slamhip_scan_select
  const double *scan_ptr = nullptr;
  size_t scan_stride = 0;
  struct ScanSlot {  // slamhip_scan_store: a filtered scan kept resident in HBM (five arrays of `cap` doubles)
    double *d = nullptr;
    int cap = 0, n = 0;
    double tot_w = 0.0;
    std::vector<double> w, f;
  };
  std::vector<ScanSlot> scan_slots;
  // slamhip_scan_filter_upload: what depends on the scanner's beam ANGLES only (cos / sin per raw beam as the scan's
  // trig provider tabulates them, the viny weighting's angular factor) is kept between scans -- a scanner's angles do
  // not change -- next to the buffers the filtered scan is assembled in
  struct ScanPrep {
    std::vector<double> angle, cos_a, sin_a, viny_f;
    int trig_mode = -1;
    double a_min = 0, a_max = 0, a_inc = 0;
    std::vector<double> r, a, c, s, w, f;
    std::vector<int> kept;
  } scan_prep;
  std::vector<double> h_weight, h_factor;  // host copies for GMapping carry-in fix-ups
  // the libm-exact modes (exact_kernels.hip): the beam ANGLES of the current scan (slamhip_scan_set_angles, or the kept
  // angles of slamhip_scan_filter_upload) -- on the host until an exact scoring call needs them in HBM --, the
  // per-pose trig tables, the reference's one GMapping cache object on the device
  std::vector<double> h_scan_angle;
  double *d_scan_angle = nullptr;
  int scan_angle_cap = 0;
  bool scan_angle_on_device = false;
  double *d_exact_trig = nullptr;
  size_t exact_trig_cap = 0;
  void *d_gm_exact_cache = nullptr;
  bool gm_exact_last = false;  // the last scoring call applied the GMapping cache itself: no host fix-up after it
  // pose / score staging
  double *d_poses = nullptr, *d_scores = nullptr, *d_pose_sc = nullptr;
  double *h_poses = nullptr, *h_scores = nullptr, *h_pose_sc = nullptr;  // pinned
  slamhip::GmPoseInfo *d_gm_info = nullptr, *h_gm_info = nullptr;
  unsigned long long *h_fprints = nullptr;  // pinned, next to h_scores
  bool want_fprints = false;      // the next score_staged() asks K1 for fingerprints (matchers' checked mode)
  int *h_pose_slot = nullptr;  // pinned: map slot of every staged pose (per-particle maps only)
  int pose_cap = 0;
  double *d_terms = nullptr;
  size_t terms_cap = 0;
  // dirty-cell staging
  int *d_dirty_xy = nullptr;
  double *d_dirty_val = nullptr;
  int dirty_cap = 0;
  // GMapping OOPE cache (gmapping_occupancy_observation_pe.h:43-44)
  int gm_cx = 0, gm_cy = 0;
  double gm_prob = -1.0;
  // low-latency completion flag (k_publish in score_kernels.hip)
  unsigned *h_done_flag = nullptr;  // pinned, coherent
  unsigned seq = 0;
  // second launch lane (stream + completion flag): the filter's two job groups score on one lane each,
  // so that one group's kernel can start while the other's is still draining / publishing
  hipStream_t stream_b = nullptr;
  unsigned *h_done_flag_b = nullptr;
  unsigned seq_b = 0;
  hipEvent_t ev_fork = nullptr;
  void *shard = nullptr;  // RCCL group of the context (shard.cpp), or null
  void *mu_scratch = nullptr, *mu_bscratch = nullptr;  // K6 work buffers (map_update.hip), owned by the context
  bool low_latency = true;
  bool stage_poses = false;  // copy poses to HBM first instead of reading them over PCIe
  // slamhip_ctx_set_option: equivalent execution paths (defaults = what is measured)
  bool filter_chains = true, k6_batch_fast = true, k6_batch_key64 = false, resident_chains = true, tbm_plane = true;
  int inert_tail = 2;  // SLAMHIP_OPT_INERT_TAIL
  int k6_path = 0;
  // profiling: event pairs recorded around scoring launches, resolved lazily in profile_read
  bool profile = false;
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;
  std::vector<char> ev_kind;  // per event pair: 0 scoring dispatch, 1 one map update (K6 pipeline)
  double prof_ms = 0.0;
  long long prof_launches = 0, prof_units = 0;
  double prof_k6_ms = 0.0;
  long long prof_k6_calls = 0, prof_k6_records = 0;
};

namespace slamhip {
// scoring target made of per-slot copy-on-write maps (tile_pool.h) instead of a bound dense window;
// pose p reads the map of slot ctx->h_pose_slot[p]
struct TiledTarget {
  const double *pool;
  const int *tables;
  int table_stride, tiles_x, width, height, origin_x, origin_y;
  double scale;
  double unknown[4];
  int nbr_ok;  // the tiles' in-tile neighbourhood masks are valid for threshold nbr_th (tile_pool.h)
  double nbr_th;
};
int ensure_pose_capacity(slamhip_ctx *ctx, int n);
// scores n poses whose (x,y,theta) sit in ctx->h_poses; results land in ctx->h_scores (synchronous)
// `off`: window of the staging buffers (poses at h_poses + 3 off, results at h_scores + off, ...);
// `async_seq` != null: return right after the launch with the number score_wait() takes (0 = the
// call was synchronous after all, nothing to wait for)
// `lane` 1: launch on the context's second stream / completion flag (zero-copy path only)
int score_staged(slamhip_ctx *ctx, int map_id, const slamhip_spe_cfg *cfg, int n_poses,
                 const TiledTarget *tiled = nullptr, int off = 0, unsigned *async_seq = nullptr, int lane = 0);
int score_wait(slamhip_ctx *ctx, unsigned seq, int lane = 0);
// orders the second lane behind everything queued on the first so far (scan upload, map updates)
int lane_fork(slamhip_ctx *ctx);
// (oie_eff, optional: a caller that passes it takes the map through its probability plane where there is one -- the
// view is then an OCC one and *oie_eff the occupancy OIE, see DeviceMap::d_prob; without it the view is the map's own)
int score_views(slamhip_ctx *ctx, int map_id, const slamhip_spe_cfg *cfg, MapView *map, ScanView *scan,
                int *cell_model, const TiledTarget *tiled = nullptr, int *oie_eff = nullptr);
int profile_event_pair(slamhip_ctx *ctx, hipEvent_t *e0, hipEvent_t *e1, int kind = 0);
// An event pair recorded AROUND a pipeline (the map update): every exit between the first record and the second --
// an empty update, a failed grow, a HIP error -- would leave a pair whose end was never recorded (or holds a stale
// time from the pool's last use), and slamhip_profile_read would fail on it.  The guard records the end on the
// way out unless close() did.
struct ProfilePairGuard {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipStream_t stream = nullptr;
  bool closed = true;
  int open(slamhip_ctx *ctx, hipStream_t st, int kind);
  int close();  // records the end of the pair; SLAMHIP_OK when profiling is off
  bool on() const { return e1 != nullptr; }
  ~ProfilePairGuard() {
    if (!closed && e1) (void)hipEventRecord(e1, stream);
  }
};

// ---- many hill-climbing chains over the GMapping OOPE in shared launches (matchers.cpp; the filter's lock-step
// replacement: one chain per particle, same map, same scan, no carry-in -- DESIGN.md section 7)
struct GmChainResult {
  double pose[3];
  double prob;
  long long calls, evaluated;
  int steps;
  int cx, cy;    // the cache entry the chain ended with
  double cprob;
  GmPoseInfo first_info;  // side outputs and raw score of the initial pose
  double first_raw;
  int error;     // 3: a one-run scan sat on the path: the caller redoes this match on the host-driven path
};
struct GmMultiChain;
// tiled + slots: chain c reads the copy-on-write map in slot slots[c] of the tile pool instead of the dense map
int gm_multi_chain_run(slamhip_ctx *ctx, GmMultiChain **scratch, int map_id, const slamhip_spe_cfg *cfg,
                       unsigned max_failed, double dt, double dr, int n, const double *inits, GmChainResult *out,
                       long long *kernels_launched, const TiledTarget *tiled = nullptr, const int *slots = nullptr);
void gm_multi_chain_free(GmMultiChain *s);
// whether gm_multi_chain_run would take n chains as ONE co-resident launch (hc_resident_gm.hip) on this context
bool gm_multi_chain_fits_resident(slamhip_ctx *ctx, int n);
}  // namespace slamhip
