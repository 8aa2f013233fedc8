// oracle/ref_pf_adapter_harness.cpp -- TEST INFRASTRUCTURE ONLY (built where /root/reference exists).
//
// Drop-in proof for the particle-filter path: HipGmappingParticleFilter
// (slam-constructor_amd/host/slamhip_gmapping_adapter.h, a LaserScanGridWorld subclass over the C-ABI)
// is compiled against the reference headers and run next to the reference's own
// GmappingParticleFilter (src/slams/gmapping/gmapping_particle_filter.h) wired like init_gmapping
// (init_gmapping.h:49-65): same seeds, same TransformedLaserScan objects through
// handle_sensor_data, both observed through World::pose() / map().
// The reference draws its seeds from std::random_device; as in ref_harness.cpp the class is shadowed
// by a queue-fed functor while the reference headers are parsed (the headers themselves are
// untouched), and `private` is opened to read per-particle poses and weights for the comparison.
// Output: oracle/_ref/libslamref_pf_adapter.so (links slam-constructor_amd/libslamhip.so).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <deque>
#include <iostream>
#include <memory>
#include <random>
#include <sstream>
#include <string>
#include <vector>

namespace slamref_seed {
static std::deque<unsigned> queue;
static unsigned next() {
  if (queue.empty()) return 12345u;
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
#include "core/maps/lazy_tiled_grid_map.h"
#include "core/maps/grid_map_scan_adders.h"
#include "core/maps/const_occupancy_estimator.h"
#include "core/scan_matchers/hill_climbing_scan_matcher.h"
#include "core/scan_matchers/weighted_mean_point_probability_spe.h"
#include "core/particle_filter.h"
#include "slams/gmapping/gmapping_grid_cell.h"
#include "slams/gmapping/gmapping_occupancy_observation_pe.h"
#include "slams/gmapping/gmapping_particle_filter.h"
#undef private
#undef protected
#undef random_device

#include "slamhip_gmapping_adapter.h"

namespace {
struct PoseLog : public WorldPoseObserver {
  std::vector<double> xs;
  void on_pose_update(const RobotPose &p) override { xs.insert(xs.end(), {p.x, p.y, p.theta}); }
};

struct Pair {
  std::shared_ptr<GridMap> ref_map;
  std::shared_ptr<GmappingParticleFilter> ref;
  slamhip_ctx *ctx = nullptr;
  std::shared_ptr<HipGmappingParticleFilter> hip;
  std::shared_ptr<PoseLog> ref_log, hip_log;
  std::deque<unsigned> hip_seeds;
  unsigned n = 0;
};
}  // namespace

extern "C" {

// gp = GMappingParams (8 doubles); seeds: n particle seeds; mode 0 = shared map (the reference's
// behaviour), 1 = per-particle maps on the HIP side (no reference counterpart: only runs)
void *refpf_create(unsigned n, int w, int h, double scale, const double *gp, const unsigned *seeds,
                   unsigned skip_rate, int mode, int strict) {
  auto *p = new Pair;
  p->n = n;
  p->ref_map = std::make_shared<UnboundedLazyTiledGridMap>(std::make_shared<GmappingBaseCell>(),
                                                           GridMapParams{w, h, scale});
  auto spe = std::make_shared<WeightedMeanPointProbabilitySPE>(
      std::make_shared<GmappingOccupancyObservationPE>(0.1, 1), std::make_shared<EvenSPW>(), skip_rate, -1.0);
  auto adder = WallDistanceBlurringScanAdder::builder()
                   .set_occupancy_estimator(std::make_shared<ConstOccupancyEstimator>(Occupancy{0.95, 1.0},
                                                                                      Occupancy{0.01, 1.0}))
                   .set_observation_quality_estimator(std::make_shared<IdleOMQE>())
                   .set_blur_distance(0.0)
                   .set_max_usable_range(std::numeric_limits<double>::infinity())
                   .build();
  auto shw = SingleStateHypothesisLSGWProperties{
      1.0, 1.0, 0, p->ref_map, std::make_shared<HillClimbingScanMatcher>(spe, 6, 0.1, 0.1), adder};
  GMappingParams gparams{gp[0], gp[1], gp[2], gp[3], gp[4], gp[5], gp[6], gp[7]};
  slamref_seed::queue.clear();
  for (unsigned i = 0; i < n; ++i) slamref_seed::queue.push_back(seeds[i]);
  p->ref = std::make_shared<GmappingParticleFilter>(shw, gparams, n);
  p->ref_log = std::make_shared<PoseLog>();
  p->ref->subscribe_pose(p->ref_log);

  if (slamhip_ctx_create(0, &p->ctx) != SLAMHIP_OK) {
    std::cerr << "refpf_create: " << slamhip_last_error() << std::endl;
    delete p;
    return nullptr;
  }
  HipGmappingParticleFilter::Config c;
  c.filter.mean_sample_xy = gp[0];
  c.filter.sigma_sample_xy = gp[1];
  c.filter.mean_sample_th = gp[2];
  c.filter.sigma_sample_th = gp[3];
  c.filter.min_sm_lim_xy = gp[4];
  c.filter.max_sm_lim_xy = gp[5];
  c.filter.min_sm_lim_th = gp[6];
  c.filter.max_sm_lim_th = gp[7];
  c.filter.hc_failed_rounds_limit = 6;
  c.filter.hc_translation = 0.1;
  c.filter.hc_rotation = 0.1;
  c.filter.sp_skip_rate = skip_rate;
  c.filter.sp_max_usable_range = -1.0;
  c.filter.oope_fullness_th = 0.1;
  c.filter.oope_window = 1;
  // strict 2 (r06): the reference's default raw trig provider and its exp bit for bit (SLAMHIP_POSE_TRIG_RAW_EXACT)
  c.filter.pose_trig = strict == 2 ? SLAMHIP_POSE_TRIG_RAW_EXACT : (strict ? SLAMHIP_POSE_TRIG_HOST : SLAMHIP_POSE_TRIG_DEVICE);
  c.adder.rule = SLAMHIP_RULE_GMAPPING;
  c.adder.scan_quality = 1.0;
  c.adder.base_occupied_prob = 0.95;
  c.adder.base_occupied_qual = 1.0;
  c.adder.base_empty_prob = 0.01;
  c.adder.base_empty_qual = 1.0;
  c.adder.blur = 0.0;
  c.adder.max_range = std::numeric_limits<double>::infinity();
  c.adder.occupancy_estimator = 0;
  c.adder.area_shift_amount = 0.0;
  c.map = GridMapParams{w, h, scale};
  c.particle_maps = mode == 1;
  c.extent_tiles = (std::max(w, h) + 127) / 128 + 1;
  c.pool_tiles = c.extent_tiles * c.extent_tiles + 32 * (int)n;
  for (unsigned i = 0; i < n; ++i) p->hip_seeds.push_back(seeds[i]);
  p->hip = std::make_shared<HipGmappingParticleFilter>(p->ctx, c, n, [p]() {
    if (p->hip_seeds.empty()) return 12345u;
    unsigned v = p->hip_seeds.front();
    p->hip_seeds.pop_front();
    return v;
  });
  p->hip_log = std::make_shared<PoseLog>();
  p->hip->subscribe_pose(p->hip_log);
  return p;
}

void refpf_destroy(void *h) {
  auto *p = static_cast<Pair *>(h);
  if (!p) return;
  p->hip.reset();
  if (p->ctx) slamhip_ctx_destroy(p->ctx);
  delete p;
}

// one handle_sensor_data on both worlds with the same scan object contents.
// out: ref_poses[3n], ref_w[n], hip_poses[3n], hip_w[n], world_pose[6] (ref pose(), hip pose()),
// flags[4] = {ref resampled, hip resampled, ref pose notifications so far, hip pose notifications}
void refpf_step(void *h, int n_pts, const double *range, const double *angle, double dx, double dy, double dth,
                unsigned resample_seed, unsigned n_extra, const unsigned *extra, double *ref_poses, double *ref_w,
                double *hip_poses, double *hip_w, double *world_pose, int *flags) {
  auto *p = static_cast<Pair *>(h);
  TransformedLaserScan ts;
  for (int i = 0; i < n_pts; ++i) ts.scan.points().push_back(ScanPoint2D::make_polar(range[i], angle[i], true));
  ts.quality = 1.0;
  ts.pose_delta = RobotPoseDelta{dx, dy, dth};
  TransformedLaserScan ts2 = ts;
  ts2.scan.trig_provider = std::make_shared<RawTrigonometryProvider>();  // the copy must not share the provider

  slamref_seed::queue.clear();
  slamref_seed::queue.push_back(resample_seed);
  for (unsigned i = 0; i < n_extra; ++i) slamref_seed::queue.push_back(extra[i]);
  const size_t before = slamref_seed::queue.size();
  p->ref->handle_sensor_data(ts);
  flags[0] = slamref_seed::queue.size() != before;
  auto &ps = p->ref->_pf.particles();
  for (size_t i = 0; i < ps.size(); ++i) {
    ref_poses[3 * i] = ps[i]->pose().x;
    ref_poses[3 * i + 1] = ps[i]->pose().y;
    ref_poses[3 * i + 2] = ps[i]->pose().theta;
    ref_w[i] = ps[i]->weight();
  }
  p->hip_seeds.clear();
  p->hip_seeds.push_back(resample_seed);
  p->hip->handle_sensor_data(ts2);
  flags[1] = p->hip->resampled_last_step();
  std::memcpy(hip_poses, p->hip->particle_poses().data(), sizeof(double) * 3 * p->n);
  std::memcpy(hip_w, p->hip->particle_weights().data(), sizeof(double) * p->n);
  const RobotPose &a = p->ref->pose(), &b = p->hip->pose();
  world_pose[0] = a.x, world_pose[1] = a.y, world_pose[2] = a.theta;
  world_pose[3] = b.x, world_pose[4] = b.y, world_pose[5] = b.theta;
  flags[2] = (int)p->ref_log->xs.size() / 3;
  flags[3] = (int)p->hip_log->xs.size() / 3;
}

// GridMap::occupancy over an external window through World::map() of the reference (0) / adapter (1)
void refpf_map_occupancy(void *h, int which, int x0, int y0, int w, int hh, double *out) {
  auto *p = static_cast<Pair *>(h);
  const GridMap &m = which ? static_cast<const LaserScanGridWorld &>(*p->hip).map()
                           : static_cast<const LaserScanGridWorld &>(*p->ref).map();
  for (int y = 0; y < hh; ++y)
    for (int x = 0; x < w; ++x) out[(size_t)y * w + x] = m.occupancy({x0 + x, y0 + y});
}

}  // extern "C"
