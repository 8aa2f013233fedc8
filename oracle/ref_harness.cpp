// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE ONLY.
//
// A thin extern "C" shim around the *unmodified* reference headers under
// /root/reference/src (slam_constructor 0.9.1).  It is compiled ONLY in the
// build container (where /root/reference is mounted) by oracle/Makefile into
// oracle/_ref/libslamref.so; no reference source is copied into this repo.
// It is used to
//   (i)  validate the C restatement in oracle/slam_oracle.c,
//   (ii) generate the golden vectors under tests/golden/ (tests/golden/make_golden.py),
//   (iii) optionally serve as the "reference" CPU baseline of bench.py when the
//        prebuilt .so travelled to the GPU box.
// Nothing in the product path (slam-constructor_amd/) may link or load it.
//
// Seeding: the reference hard-wires std::random_device in
//   src/core/particle_filter.h:51-52, src/slams/gmapping/gmapping_world.h:51.
// We shadow it with a FIFO of injected seeds without editing
// the reference.  Access control is relaxed (#define private public) only so the
// harness can *read* cell payloads (GmappingBaseCell::obst) and particle state.

#include <algorithm>
#include <array>
#include <cassert>
#include <cmath>
#include <cstring>
#include <deque>
#include <fstream>
#include <functional>
#include <iomanip>
#include <iostream>
#include <iterator>
#include <limits>
#include <map>
#include <memory>
#include <random>
#include <sstream>
#include <string>
#include <tuple>
#include <unordered_set>
#include <utility>
#include <vector>

namespace slamref_seed {
static std::deque<unsigned> queue;
static unsigned fallback = 12345u;
static unsigned next() {
  if (queue.empty()) return fallback;
  unsigned v = queue.front();
  queue.pop_front();
  return v;
}
}  // namespace slamref_seed

namespace std {
struct slamref_fixed_random_device {
  using result_type = unsigned;
  slamref_fixed_random_device() {}
  unsigned operator()() { return slamref_seed::next(); }
  static constexpr unsigned min() { return 0; }
  static constexpr unsigned max() { return 0xffffffffu; }
};
}  // namespace std

#define random_device slamref_fixed_random_device
#define private public
#define protected public

#include "core/maps/plain_grid_map.h"
#include "core/maps/lazy_tiled_grid_map.h"
#include "core/maps/naive_grid_cells.h"
#include "core/maps/tbm_grid_cells.h"
#include "core/maps/grid_map_scan_adders.h"
#include "core/maps/const_occupancy_estimator.h"
#include "core/maps/area_occupancy_estimator.h"
#include "core/scan_matchers/observation_impact_estimators.h"
#include "core/scan_matchers/occupancy_observation_probability.h"
#include "core/scan_matchers/weighted_mean_point_probability_spe.h"
#include "core/scan_matchers/monte_carlo_scan_matcher.h"
#include "core/scan_matchers/hill_climbing_scan_matcher.h"
#include "core/scan_matchers/brute_force_scan_matcher.h"
#include "core/particle_filter.h"
#include "slams/gmapping/gmapping_grid_cell.h"
#include "slams/gmapping/gmapping_occupancy_observation_pe.h"
#include "slams/gmapping/gmapping_particle_filter.h"
#include "utils/data_generation/map_primitives.h"
#include "utils/data_generation/grid_map_patcher.h"
#include "utils/data_generation/laser_scan_generator.h"
#include "utils/map_dumpers.h"
#include "../test/core/mock_grid_cell.h"

#undef private
#undef protected
#undef random_device

// ---------------------------------------------------------------------------
// enums shared with oracle/slam_oracle.h and include/slamhip.h (same values)
enum { CELL_MEAN = 0, CELL_TBM = 1, CELL_GMAPPING = 2, CELL_AFFINE = 3, CELL_MOCK = 4 };
enum { MAP_PLAIN = 0, MAP_UNBOUNDED_PLAIN = 1, MAP_LAZY_TILED = 2, MAP_UNBOUNDED_LAZY_TILED = 3 };
enum { OOPE_OBSTACLE = 0, OOPE_MAX = 1, OOPE_MEAN = 2, OOPE_OVERLAP = 3, OOPE_GMAPPING = 4 };
enum { OIE_DISCREPANCY = 0, OIE_OCCUPANCY = 1 };
enum { SPW_EVEN = 0, SPW_VINY = 1, SPW_AHR = 2 };
enum { TRIG_RAW = 0, TRIG_CACHED = 1 };
enum { SM_MC = 0, SM_HC = 1, SM_BF = 2 };

struct RefMap {
  std::shared_ptr<GridMap> map;
  int cell_model;
};

struct RefScan {
  LaserScan2D scan;
};

struct RefSpe {
  std::shared_ptr<WeightedMeanPointProbabilitySPE> spe;
  std::shared_ptr<ScanPointWeighting> spw;
  std::shared_ptr<OccupancyObservationProbabilityEstimator> oope;
};

struct RefMatcher {
  std::shared_ptr<GridScanMatcher> sm;
};

struct TraceObserver : public GridScanMatcherObserver {
  std::vector<double> poses;   // 3 per test
  std::vector<double> scores;  // 1 per test
  std::vector<int> accepted;   // 1 per test
  void on_scan_test(const RobotPose &p, const LaserScan2D &, double s) override {
    poses.push_back(p.x);
    poses.push_back(p.y);
    poses.push_back(p.theta);
    scores.push_back(s);
    accepted.push_back(0);
  }
  void on_pose_update(const RobotPose &, const LaserScan2D &, double) override {
    if (!accepted.empty()) accepted.back() = 1;
  }
};

static std::shared_ptr<GridCell> make_cell(int model, double mock_prob) {
  switch (model) {
    case CELL_MEAN: return std::make_shared<MeanProbabilityCell>();
    case CELL_TBM: return std::make_shared<TbmOccConsistentCell>();
    case CELL_GMAPPING: return std::make_shared<GmappingBaseCell>();
    case CELL_AFFINE: return std::make_shared<AffineQualityMergeCell>();
    case CELL_MOCK: return std::make_shared<MockGridCell>(mock_prob);
  }
  return nullptr;
}

static std::shared_ptr<ObservationImpactEstimator> make_oie(int k) {
  if (k == OIE_OCCUPANCY) return std::make_shared<OccupancyOIE>();
  return std::make_shared<DiscrepancyOIE>();
}

static std::shared_ptr<OccupancyObservationProbabilityEstimator> make_oope(
    int kind, int oie, double gm_th, unsigned gm_win) {
  auto o = make_oie(oie);
  switch (kind) {
    case OOPE_OBSTACLE: return std::make_shared<ObstacleBasedOccupancyObservationPE>(o);
    case OOPE_MAX: return std::make_shared<MaxOccupancyObservationPE>(o);
    case OOPE_MEAN: return std::make_shared<MeanOccupancyObservationPE>(o);
    case OOPE_OVERLAP: return std::make_shared<OverlapWeightedOccupancyObservationPE>(o);
    case OOPE_GMAPPING: return std::make_shared<GmappingOccupancyObservationPE>(gm_th, gm_win);
  }
  return nullptr;
}

static int payload_stride(int model) {
  return model == CELL_TBM ? 4 : (model == CELL_GMAPPING ? 3 : 1);
}

static void cell_payload(const GridCell &c, int model, double *out) {
  if (model == CELL_TBM) {
    const auto &t = static_cast<const TbmBaseCell &>(c).belief();
    out[0] = t.unknown();
    out[1] = t.empty();
    out[2] = t.occupied();
    out[3] = t.conflict();
  } else if (model == CELL_GMAPPING) {
    const auto &g = static_cast<const GmappingBaseCell &>(c);
    out[0] = g.occupancy().prob_occ;
    out[1] = g.obst.x;
    out[2] = g.obst.y;
  } else {
    out[0] = c.occupancy().prob_occ;
  }
}

extern "C" {

// ---- seeds ---------------------------------------------------------------
void ref_seed_push(unsigned s) { slamref_seed::queue.push_back(s); }
void ref_seed_clear() { slamref_seed::queue.clear(); }
int ref_seed_pending() { return (int)slamref_seed::queue.size(); }

// ---- maps ----------------------------------------------------------------
void *ref_map_create(int cell_model, int map_type, int w, int h, double scale,
                     double mock_prob) {
  auto proto = make_cell(cell_model, mock_prob);
  if (!proto) return nullptr;
  GridMapParams p{w, h, scale};
  auto *m = new RefMap;
  m->cell_model = cell_model;
  switch (map_type) {
    case MAP_PLAIN: m->map = std::make_shared<PlainGridMap>(proto, p); break;
    case MAP_UNBOUNDED_PLAIN: m->map = std::make_shared<UnboundedPlainGridMap>(proto, p); break;
    case MAP_LAZY_TILED: m->map = std::make_shared<LazyTiledGridMap>(proto, p); break;
    case MAP_UNBOUNDED_LAZY_TILED:
      m->map = std::make_shared<UnboundedLazyTiledGridMap>(proto, p);
      break;
    default: delete m; return nullptr;
  }
  return m;
}
void ref_map_destroy(void *h) { delete static_cast<RefMap *>(h); }

// `GridMap copy = original;` of the tiled maps: the copy shares every tile with the original until one
// of them writes (lazy_tiled_grid_map.h:40-71) -- what `*new_particle = *sampled` does to a particle's map
// (particle_filter.h:92-96).  Null for map classes without those semantics.
void *ref_map_copy(void *h) {
  auto *src = static_cast<RefMap *>(h);
  auto *m = new RefMap;
  m->cell_model = src->cell_model;
  if (auto u = std::dynamic_pointer_cast<UnboundedLazyTiledGridMap>(src->map))
    m->map = std::make_shared<UnboundedLazyTiledGridMap>(*u);
  else if (auto l = std::dynamic_pointer_cast<LazyTiledGridMap>(src->map))
    m->map = std::make_shared<LazyTiledGridMap>(*l);
  else {
    delete m;
    return nullptr;
  }
  return m;
}

// geometry: width, height, origin_x, origin_y, + scale
void ref_map_geometry(void *h, int *out4, double *scale) {
  auto &m = *static_cast<RefMap *>(h)->map;
  out4[0] = m.width();
  out4[1] = m.height();
  out4[2] = m.origin().x;
  out4[3] = m.origin().y;
  *scale = m.scale();
}

void ref_map_update(void *h, int x, int y, int is_occ, double prob, double qual,
                    double obst_x, double obst_y, double quality) {
  auto &m = *static_cast<RefMap *>(h)->map;
  m.update({x, y}, AreaOccupancyObservation{bool(is_occ), {prob, qual}, {obst_x, obst_y}, quality});
}

// text raster through the reference GridMapPatcher (explicit offset form)
void ref_map_stamp_text(void *h, const char *text, int off_x, int off_y, int w_zoom,
                        int h_zoom) {
  auto &m = *static_cast<RefMap *>(h)->map;
  std::stringstream ss{std::string(text)};
  GridMapPatcher{}.apply_text_raster(m, ss, DiscretePoint2D{off_x, off_y}, w_zoom, h_zoom);
}

// cecum primitive text (bnd_pos: 0 Left, 1 Right, 2 Top, 3 Bot); returns length
int ref_cecum_text(int w, int h, int bnd_pos, char *out, int cap) {
  using C = CecumTextRasterMapPrimitive;
  C c{w, h, static_cast<C::BoundPosition>(bnd_pos)};
  std::string s = c.text_raster();
  if ((int)s.size() + 1 > cap) return -(int)s.size() - 1;
  std::memcpy(out, s.c_str(), s.size() + 1);
  return (int)s.size();
}

// payload of external cells [x0, x0+w) x [y0, y0+h), row-major [y][x][stride]
void ref_map_export(void *h, int x0, int y0, int w, int hh, double *out) {
  auto *rm = static_cast<RefMap *>(h);
  int st = payload_stride(rm->cell_model);
  for (int y = 0; y < hh; ++y)
    for (int x = 0; x < w; ++x)
      cell_payload((*rm->map)[{x0 + x, y0 + y}], rm->cell_model,
                   out + (size_t(y) * w + x) * st);
}

// whole internal window (external = internal - origin)
void ref_map_export_all(void *h, double *out) {
  auto *rm = static_cast<RefMap *>(h);
  auto &m = *rm->map;
  ref_map_export(h, -m.origin().x, -m.origin().y, m.width(), m.height(), out);
}

// per-cell update state that is not part of the scoring payload: MeanProbabilityCell::_n
// (naive_grid_cells.h:43) -> 1 double; GmappingBaseCell::_hits/_tries (gmapping_grid_cell.h:41) -> 2
int ref_map_export_aux(void *h, double *out) {
  auto *rm = static_cast<RefMap *>(h);
  auto &m = *rm->map;
  const int w = m.width(), hh = m.height(), ox = m.origin().x, oy = m.origin().y;
  if (rm->cell_model != CELL_MEAN && rm->cell_model != CELL_GMAPPING) return 0;
  const int st = rm->cell_model == CELL_MEAN ? 1 : 2;
  for (int y = 0; y < hh; ++y)
    for (int x = 0; x < w; ++x) {
      const GridCell &c = m[{x - ox, y - oy}];
      double *o = out + (size_t(y) * w + x) * st;
      if (rm->cell_model == CELL_MEAN) {
        o[0] = static_cast<const MeanProbabilityCell &>(c)._n;
      } else {
        o[0] = static_cast<const GmappingBaseCell &>(c)._hits;
        o[1] = static_cast<const GmappingBaseCell &>(c)._tries;
      }
    }
  return st;
}

// bulk GridMap::update of n external cells with {occupied, {prob, 1}, (0,0), quality 1}.  On an
// AffineQualityMergeCell map this sets prob_occ to exactly `prob` ((1-1)*old + 1*new), which lets
// bench.py rebuild its synthetic occupancy map as a reference map for the "reference" CPU baseline.
void ref_map_update_bulk(void *h, int n, const int *xy, const double *prob) {
  auto &m = *static_cast<RefMap *>(h)->map;
  for (int i = 0; i < n; ++i)
    m.update({xy[2 * i], xy[2 * i + 1]}, AreaOccupancyObservation{true, {prob[i], 1.0}, {0, 0}, 1.0});
}

// GridMapToPgmDumber::dump_map (src/utils/map_dumpers.h:65-90) of the map to `path`
int ref_map_dump_pgm(void *h, const char *path) {
  auto &m = *static_cast<RefMap *>(h)->map;
  std::ofstream f(path, std::ios::binary | std::ios::out);
  GridMapToPgmDumber::dump_map(f, m);
  return f.good() ? 0 : -1;
}

// GridMap::save_state (plain_grid_map.h:79-99) to a file: the `.map` fixture format of
// lslam2D_bag_runner's dump_state / sm_runner (src/utils/sm_runner.cpp:36-50)

int ref_map_save_state(void *h, const char *path) {
  auto &m = *static_cast<RefMap *>(h)->map;
  auto buf = m.save_state();
  std::ofstream f(path, std::ios::binary);
  f.write(buf.data(), buf.size());
  return (int)buf.size();
}

void ref_map_unknown_payload(void *h, double *out) {
  auto *rm = static_cast<RefMap *>(h);
  auto c = rm->map->new_cell();
  cell_payload(*c, rm->cell_model, out);
}

// ---- scans ---------------------------------------------------------------
void *ref_scan_create(int n, const double *ranges, const double *angles, const int *is_occ,
                      int trig_mode, double a_min, double a_max, double a_inc) {
  auto *s = new RefScan;
  s->scan.points().reserve(n);
  for (int i = 0; i < n; ++i)
    s->scan.points().emplace_back(ranges[i], angles[i], is_occ ? bool(is_occ[i]) : true);
  if (trig_mode == TRIG_CACHED) {
    auto p = std::make_shared<CachedTrigonometryProvider>();
    p->update(a_min, a_max, a_inc);  // caller passes a_max the way laser_scan_observer.h:80 does
    s->scan.trig_provider = p;
  } else {
    s->scan.trig_provider = std::make_shared<RawTrigonometryProvider>();
  }
  return s;
}
void ref_scan_destroy(void *h) { delete static_cast<RefScan *>(h); }
int ref_scan_size(void *h) { return (int)static_cast<RefScan *>(h)->scan.points().size(); }
void ref_scan_get(void *h, double *ranges, double *angles, int *is_occ, double *factor) {
  auto &pts = static_cast<RefScan *>(h)->scan.points();
  for (size_t i = 0; i < pts.size(); ++i) {
    ranges[i] = pts[i].range();
    angles[i] = pts[i].angle();
    if (is_occ) is_occ[i] = pts[i].is_occupied();
    if (factor) factor[i] = pts[i].factor();
  }
}
void ref_scan_set_factor(void *h, int i, double f) {
  static_cast<RefScan *>(h)->scan.points()[i].set_factor(f);
}
// cached-provider table (n entries written; returns n) -- trigonometry_utils.h:62-78
int ref_scan_trig_table(void *h, double *sin_out, double *cos_out, int cap) {
  auto p = std::dynamic_pointer_cast<CachedTrigonometryProvider>(
      static_cast<RefScan *>(h)->scan.trig_provider);
  if (!p) return 0;
  int n = (int)p->_sin.size();
  for (int i = 0; i < n && i < cap; ++i) {
    sin_out[i] = p->_sin[i];
    cos_out[i] = p->_cos[i];
  }
  return n;
}

// LaserScanGenerator (utils/data_generation/laser_scan_generator.h:35-80)
void *ref_scan_generate(void *map_h, double x, double y, double th, double max_dist,
                        double fov_deg, unsigned pts_nm, double occ_threshold) {
  auto &m = *static_cast<RefMap *>(map_h)->map;
  auto *s = new RefScan;
  s->scan = LaserScanGenerator{to_lsp(max_dist, fov_deg, pts_nm)}.laser_scan_2D(
      m, RobotPose{x, y, th}, occ_threshold);
  return s;
}

// ---- scan probability estimator --------------------------------------------
void *ref_spe_create(int oope_kind, int oie_kind, int weighting, unsigned skip_rate,
                     double max_range, double gm_fullness_th, unsigned gm_window) {
  auto *s = new RefSpe;
  s->oope = make_oope(oope_kind, oie_kind, gm_fullness_th, gm_window);
  switch (weighting) {
    case SPW_VINY: s->spw = std::make_shared<VinySlamSPW>(); break;
    case SPW_AHR: s->spw = std::make_shared<AngleHistogramReciprocalSPW>(); break;
    default: s->spw = std::make_shared<EvenSPW>(); break;
  }
  s->spe = std::make_shared<WeightedMeanPointProbabilitySPE>(s->oope, s->spw, skip_rate,
                                                             max_range);
  return s;
}
void ref_spe_destroy(void *h) { delete static_cast<RefSpe *>(h); }

void *ref_filter_scan(void *spe_h, void *scan_h, double x, double y, double th, void *map_h) {
  auto &spe = *static_cast<RefSpe *>(spe_h)->spe;
  auto &scan = static_cast<RefScan *>(scan_h)->scan;
  auto &m = *static_cast<RefMap *>(map_h)->map;
  auto *out = new RefScan;
  out->scan = spe.filter_scan(scan, RobotPose{x, y, th}, m);
  return out;
}

// weights as the scorer sees them (call after ref_filter_scan on that scan)
void ref_scan_weights(void *spe_h, void *scan_h, double *out) {
  auto &spw = *static_cast<RefSpe *>(spe_h)->spw;
  auto &pts = static_cast<RefScan *>(scan_h)->scan.points();
  for (size_t i = 0; i < pts.size(); ++i) out[i] = spw.weight(pts, i);
}

// estimate_scan_probability for n poses; area = (bot, top, left, right) of
// SPEParams::sp_analysis_area (all zero = point)
void ref_score(void *spe_h, void *scan_h, void *map_h, int n, const double *poses,
               const double *area4, double *out) {
  auto &spe = *static_cast<RefSpe *>(spe_h)->spe;
  auto &scan = static_cast<RefScan *>(scan_h)->scan;
  auto &m = *static_cast<RefMap *>(map_h)->map;
  ScanProbabilityEstimator::SPEParams prm;
  if (area4) prm.sp_analysis_area = LightWeightRectangle{area4[0], area4[1], area4[2], area4[3]};
  for (int i = 0; i < n; ++i)
    out[i] = spe.estimate_scan_probability(
        scan, RobotPose{poses[3 * i], poses[3 * i + 1], poses[3 * i + 2]}, m, prm);
}

// single OOPE probability (occupancy_observation_probability_test.cpp:37-44 shape)
double ref_oope_probability(int oope_kind, int oie_kind, void *map_h, double ox, double oy,
                            const double *range4) {
  auto &m = *static_cast<RefMap *>(map_h)->map;
  auto oope = make_oope(oope_kind, oie_kind, 0.1, 1);
  LightWeightRectangle r{range4[0], range4[1], range4[2], range4[3]};
  Point2D obst{ox, oy};
  return oope->probability(AreaOccupancyObservation{true, {1, 1}, obst, 1},
                           r.move_center(obst), m);
}

// ---- scan matchers -----------------------------------------------------------
// kind SM_MC: p = {seed, sigma_t, sigma_r, failed_limit, attempts_limit}
// kind SM_HC: p = {failed_rounds_limit, d_t, d_r}
// kind SM_BF: p = {from_x,to_x,step_x, from_y,to_y,step_y, from_t,to_t,step_t}
void *ref_matcher_create(int kind, void *spe_h, const double *p) {
  auto spe = static_cast<RefSpe *>(spe_h)->spe;
  auto *m = new RefMatcher;
  switch (kind) {
    case SM_MC:
      m->sm = std::make_shared<MonteCarloScanMatcher>(spe, (unsigned)p[0], p[1], p[2],
                                                      (unsigned)p[3], (unsigned)p[4]);
      break;
    case SM_HC:
      m->sm = std::make_shared<HillClimbingScanMatcher>(spe, (unsigned)p[0], p[1], p[2]);
      break;
    case SM_BF:
      m->sm = std::make_shared<BruteForceScanMatcher>(spe, p[0], p[1], p[2], p[3], p[4], p[5],
                                                      p[6], p[7], p[8]);
      break;
    default: delete m; return nullptr;
  }
  return m;
}
void ref_matcher_destroy(void *h) { delete static_cast<RefMatcher *>(h); }
void ref_matcher_reset_state(void *h) { static_cast<RefMatcher *>(h)->sm->reset_state(); }

// process_scan with a trace observer.  Returns number of scorer calls (on_scan_test
// events); writes up to cap of them.  out_res = {best_prob, dx, dy, dth}.
int ref_process_scan(void *m_h, void *scan_h, double x, double y, double th, void *map_h,
                     double *out_res, int cap, double *tr_poses, double *tr_scores,
                     int *tr_accepted, int *filtered_n) {
  auto &sm = *static_cast<RefMatcher *>(m_h)->sm;
  auto &map = *static_cast<RefMap *>(map_h)->map;
  TransformedLaserScan ts;
  ts.scan = static_cast<RefScan *>(scan_h)->scan;
  ts.quality = 1.0;
  auto obs = std::make_shared<TraceObserver>();
  sm.subscribe(obs);
  RobotPoseDelta d;
  double prob = sm.process_scan(ts, RobotPose{x, y, th}, map, d);
  sm.unsubscribe(obs);
  out_res[0] = prob;
  out_res[1] = d.x;
  out_res[2] = d.y;
  out_res[3] = d.theta;
  int n = (int)obs->scores.size();
  for (int i = 0; i < n && i < cap; ++i) {
    if (tr_poses) {
      tr_poses[3 * i] = obs->poses[3 * i];
      tr_poses[3 * i + 1] = obs->poses[3 * i + 1];
      tr_poses[3 * i + 2] = obs->poses[3 * i + 2];
    }
    if (tr_scores) tr_scores[i] = obs->scores[i];
    if (tr_accepted) tr_accepted[i] = obs->accepted[i];
  }
  if (filtered_n) {
    auto f = sm.filter_scan(ts.scan, RobotPose{x, y, th}, map);
    *filtered_n = (int)f.points().size();
  }
  return n;
}

// Pose enumerator known answers.  kind SM_MC / SM_HC, all rejected.
int ref_enumerate_all_rejected(int kind, const double *p, double x, double y, double th,
                               int cap, double *out_poses) {
  std::shared_ptr<PoseEnumerator> pe;
  if (kind == SM_MC)
    pe = std::make_shared<GaussianPoseEnumerator>((unsigned)p[0], p[1], p[2], (unsigned)p[3],
                                                  (unsigned)p[4]);
  else
    pe = std::make_shared<FailedRoundsLimitedPoseEnumerator<Distorsion1DPoseEnumerator>>(
        (unsigned)p[0], p[1], p[2]);
  RobotPose base{x, y, th};
  int n = 0;
  while (pe->has_next() && n < cap) {
    auto q = pe->next(base);
    out_poses[3 * n] = q.x;
    out_poses[3 * n + 1] = q.y;
    out_poses[3 * n + 2] = q.theta;
    ++n;
    pe->feedback(false);
  }
  return n;
}

// ---- map update (append_scan) -------------------------------------------------
// occ_est: 0 const, 1 area.  base = {occ_prob, occ_qual, empty_prob, empty_qual}
void ref_append_scan(void *map_h, void *scan_h, double x, double y, double th, double quality,
                     int occ_est, const double *base4, double blur, double max_range) {
  auto &m = *static_cast<RefMap *>(map_h)->map;
  auto &scan = static_cast<RefScan *>(scan_h)->scan;
  std::shared_ptr<CellOccupancyEstimator> est;
  Occupancy bo{base4[0], base4[1]}, be{base4[2], base4[3]};
  if (occ_est == 1)
    est = std::make_shared<AreaOccupancyEstimator>(bo, be);
  else
    est = std::make_shared<ConstOccupancyEstimator>(bo, be);
  auto adder = WallDistanceBlurringScanAdder::builder()
                   .set_occupancy_estimator(est)
                   .set_observation_quality_estimator(std::make_shared<IdleOMQE>())
                   .set_blur_distance(blur)
                   .set_max_usable_range(max_range)
                   .build();
  adder->append_scan(m, RobotPose{x, y, th}, scan, quality, 0);
}

// the same with the observation-mapping-quality estimator chosen: omqe 0 = IdleOMQE, 1 = AngleHistogramResiprocalOMQE
// (grid_map_scan_adders.h:24-43; what init_omqe builds for slam/mapping/observation_quality_estimator/typetype =
// idle / ahr, init_occupancy_mapping.h:64-80)
void ref_append_scan_omqe(void *map_h, void *scan_h, double x, double y, double th, double quality,
                          int occ_est, const double *base4, double blur, double max_range, int omqe) {
  auto &m = *static_cast<RefMap *>(map_h)->map;
  auto &scan = static_cast<RefScan *>(scan_h)->scan;
  std::shared_ptr<CellOccupancyEstimator> est;
  Occupancy bo{base4[0], base4[1]}, be{base4[2], base4[3]};
  if (occ_est == 1)
    est = std::make_shared<AreaOccupancyEstimator>(bo, be);
  else
    est = std::make_shared<ConstOccupancyEstimator>(bo, be);
  std::shared_ptr<ObservationMappingQualityEstimator> q;
  if (omqe == 1) q = std::make_shared<AngleHistogramResiprocalOMQE>();
  else q = std::make_shared<IdleOMQE>();
  auto adder = WallDistanceBlurringScanAdder::builder()
                   .set_occupancy_estimator(est)
                   .set_observation_quality_estimator(q)
                   .set_blur_distance(blur)
                   .set_max_usable_range(max_range)
                   .build();
  adder->append_scan(m, RobotPose{x, y, th}, scan, quality, 0);
}

// AngleHistogramResiprocalOMQE::quality of every point of the scan (after reset(scan))
void ref_omqe_quality(void *scan_h, double *out) {
  auto &scan = static_cast<RefScan *>(scan_h)->scan;
  AngleHistogramResiprocalOMQE q;
  q.reset(scan);
  const auto &pts = scan.points();
  for (size_t i = 0; i < pts.size(); ++i) out[i] = q.quality(pts, i);
}

// world_to_cells of a segment (regular_squares_grid.h:56-101); returns count
int ref_world_to_cells(void *map_h, double x0, double y0, double x1, double y1, int cap,
                       int *out_xy) {
  auto &m = *static_cast<RefMap *>(map_h)->map;
  auto cells = m.world_to_cells(Segment2D{{x0, y0}, {x1, y1}});
  int n = (int)cells.size();
  for (int i = 0; i < n && i < cap; ++i) {
    out_xy[2 * i] = cells[i].x;
    out_xy[2 * i + 1] = cells[i].y;
  }
  return n;
}

// ---- particle filter pieces ------------------------------------------------------
struct WParticle : public Particle {};

// UniformResamling on raw weights (particle_filter.h:34-66).  seed injected.
// returns resampling_is_required; writes indices (always computed).
int ref_resample(int n, const double *weights, unsigned seed, unsigned *out_idx) {
  std::vector<std::shared_ptr<WParticle>> ps;
  for (int i = 0; i < n; ++i) {
    auto p = std::make_shared<WParticle>();
    p->set_weight(weights[i]);
    ps.push_back(p);
  }
  UniformResamling<std::shared_ptr<WParticle>> r;
  int req = r.resampling_is_required(ps);
  slamref_seed::queue.push_front(seed);
  auto idx = r.resample(ps);
  for (int i = 0; i < n; ++i) out_idx[i] = idx[i];
  return req;
}

// ---- GMapping particle filter ------------------------------------------------------
struct RefGmapping {
  std::shared_ptr<GridMap> map;
  std::shared_ptr<GmappingParticleFilter> pf;
  std::shared_ptr<WeightedMeanPointProbabilitySPE> spe;
  int cell_model;
};

// Mirrors init_gmapping (slams/gmapping/init_gmapping.h:49-65) with explicit params.
// gp = {mean_xy, sigma_xy, mean_th, sigma_th, min_lim_xy, max_lim_xy, min_lim_th, max_lim_th}
// seeds: n particle seeds (consumed one per GmappingWorld ctor, gmapping_world.h:51)
void *ref_gmapping_create(unsigned n, int w, int h, double scale, const double *gp,
                          const unsigned *seeds, unsigned skip_rate, double max_range,
                          int occ_est, const double *base4, double blur, double map_max_range,
                          unsigned hc_limit, double hc_dt, double hc_dr) {
  auto *g = new RefGmapping;
  g->cell_model = CELL_GMAPPING;
  g->map = std::make_shared<UnboundedLazyTiledGridMap>(std::make_shared<GmappingBaseCell>(),
                                                       GridMapParams{w, h, scale});
  auto oope = std::make_shared<GmappingOccupancyObservationPE>(0.1, 1);
  g->spe = std::make_shared<WeightedMeanPointProbabilitySPE>(
      oope, std::make_shared<EvenSPW>(), skip_rate, max_range);
  std::shared_ptr<CellOccupancyEstimator> est;
  Occupancy bo{base4[0], base4[1]}, be{base4[2], base4[3]};
  if (occ_est == 1)
    est = std::make_shared<AreaOccupancyEstimator>(bo, be);
  else
    est = std::make_shared<ConstOccupancyEstimator>(bo, be);
  auto adder = WallDistanceBlurringScanAdder::builder()
                   .set_occupancy_estimator(est)
                   .set_observation_quality_estimator(std::make_shared<IdleOMQE>())
                   .set_blur_distance(blur)
                   .set_max_usable_range(map_max_range)
                   .build();
  auto shw = SingleStateHypothesisLSGWProperties{
      1.0, 1.0, 0, g->map,
      std::make_shared<HillClimbingScanMatcher>(g->spe, hc_limit, hc_dt, hc_dr), adder};
  GMappingParams gparams{gp[0], gp[1], gp[2], gp[3], gp[4], gp[5], gp[6], gp[7]};
  slamref_seed::queue.clear();
  for (unsigned i = 0; i < n; ++i) slamref_seed::queue.push_back(seeds[i]);
  g->pf = std::make_shared<GmappingParticleFilter>(shw, gparams, n);
  return g;
}
void ref_gmapping_destroy(void *h) { delete static_cast<RefGmapping *>(h); }

// returns a non-owning RefMap view of the shared map (caller frees with ref_map_destroy)
void *ref_gmapping_map(void *h) {
  auto *g = static_cast<RefGmapping *>(h);
  auto *m = new RefMap;
  m->map = g->map;
  m->cell_model = CELL_GMAPPING;
  return m;
}

// One handle_sensor_data step.  resample_seed is consumed only if resampling happens
// (particle_filter.h:51-52), extra_seeds feed GmappingWorld ctors of duplicated particles
// (create_particle in particle_filter.h:94).  Outputs per particle AFTER the step:
//   poses[3n], weights[n], is_master[n]; out_flags = {resampled}
void ref_gmapping_step(void *h, void *scan_h, double dx, double dy, double dth,
                       unsigned resample_seed, unsigned n_extra, const unsigned *extra_seeds,
                       double *poses, double *weights, int *is_master, int *out_flags) {
  auto *g = static_cast<RefGmapping *>(h);
  TransformedLaserScan ts;
  ts.scan = static_cast<RefScan *>(scan_h)->scan;
  ts.quality = 1.0;
  ts.pose_delta = RobotPoseDelta{dx, dy, dth};
  slamref_seed::queue.clear();
  slamref_seed::queue.push_back(resample_seed);
  for (unsigned i = 0; i < n_extra; ++i) slamref_seed::queue.push_back(extra_seeds[i]);
  size_t before = slamref_seed::queue.size();
  g->pf->handle_sensor_data(ts);
  out_flags[0] = slamref_seed::queue.size() != before;
  auto &ps = g->pf->_pf.particles();
  for (size_t i = 0; i < ps.size(); ++i) {
    poses[3 * i] = ps[i]->pose().x;
    poses[3 * i + 1] = ps[i]->pose().y;
    poses[3 * i + 2] = ps[i]->pose().theta;
    weights[i] = ps[i]->weight();
    is_master[i] = ps[i]->is_master();
  }
}

// per-particle scan-matching gate state (gmapping_world.h:74-77,116-119)
void ref_gmapping_gate(void *h, double *delta_since, double *next_delta) {
  auto *g = static_cast<RefGmapping *>(h);
  auto &ps = g->pf->_pf.particles();
  for (size_t i = 0; i < ps.size(); ++i) {
    delta_since[3 * i] = ps[i]->_delta_since_last_sm.x;
    delta_since[3 * i + 1] = ps[i]->_delta_since_last_sm.y;
    delta_since[3 * i + 2] = ps[i]->_delta_since_last_sm.theta;
    next_delta[3 * i] = ps[i]->_next_sm_delta.x;
    next_delta[3 * i + 1] = ps[i]->_next_sm_delta.y;
    next_delta[3 * i + 2] = ps[i]->_next_sm_delta.theta;
  }
}

}  // extern "C"
