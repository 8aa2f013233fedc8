// oracle/ref_world_harness.cpp -- TEST INFRASTRUCTURE ONLY (built where /root/reference exists).
//
// Drop-in proof over a multi-scan loop: the reference's own single-hypothesis world
// (SingleStateHypothesisLaserScanGridWorld built by init_1h_slam, src/utils/init_slam.h:12-25) and the
// same world built by init_hip_1h_slam (slam-constructor_amd/host/slamhip_init_slam.h: reference world,
// map and scan adder, HIP scan matcher) are fed the same scans.  After every scan the reference scan
// adder updates the host map (single_state_hypothesis_laser_scan_grid_world.h:52-65), so the HIP
// matcher has to see those updates in HBM before its next match; the pose after every scan and the
// final maps are compared.
// Output: oracle/_ref/libslamref_world.so (links slam-constructor_amd/libslamhip.so).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "utils/init_slam.h"
#include "utils/data_generation/map_primitives.h"
#include "utils/data_generation/grid_map_patcher.h"
#include "utils/data_generation/laser_scan_generator.h"
#include "../test/core/mock_grid_cell.h"

#include "slamhip_init_slam.h"
#include "slamhip_resident_world.h"

namespace {

// preset 0: config/slams/tiny_slam_base.properties (+ tiny_mean_cell, monte_carlo_scan_matching)
// preset 1: config/slams/viny_slam_base.properties
// presets 2, 3: the same two with the `ahr` observation quality estimator (AngleHistogramResiprocalOMQE,
// init_occupancy_mapping.h:64-80 -- the key really is ".../typetype")
// presets 4, 5: presets 0, 1 with slam/occupancy_estimator/type = area (init_occupancy_mapping.h:38-62)
void fill_props(MapPropertiesProvider &p, int preset, int matcher, unsigned seed, int strict, double size_m) {
  auto S = [&](const char *k, const std::string &v) { p.set_property(k, v); };
  const bool area = preset >= 4;  // presets 4, 5: tinySLAM / vinySLAM with the AreaOccupancyEstimator (see below)
  if (area) preset -= 4;
  if (preset >= 2) {
    S("slam/mapping/observation_quality_estimator/typetype", "ahr");
    preset -= 2;
  }
  if (preset == 0) {
    S("slam/mapping/blur", "0.5");
    S("slam/occupancy_estimator/type", "const");
    S("slam/occupancy_estimator/base_occupied/prob", "0.95");
    S("slam/occupancy_estimator/base_empty/prob", "0.01");
    S("slam/mapping/grid/area/type", "mean_probability");
    S("slam/mapping/corrected_pose_quality", "0.9");
    S("slam/mapping/raw_pose_quality", "0.6");
    S("slam/scmtch/spe/wmpp/weighting/type", "even");
  } else {
    S("slam/mapping/blur", "0.3");
    S("slam/occupancy_estimator/type", "const");
    S("slam/occupancy_estimator/base_occupied/prob", "0.95");
    S("slam/occupancy_estimator/base_occupied/qual", "0.04");
    S("slam/occupancy_estimator/base_empty/prob", "0.01");
    S("slam/occupancy_estimator/base_empty/qual", "0.003");
    S("slam/mapping/grid/area/type", "tbm_consistent");
    S("slam/mapping/corrected_pose_quality", "0.9");
    S("slam/mapping/raw_pose_quality", "0.6");
    S("slam/scmtch/spe/wmpp/weighting/type", "viny");
  }
  // (the occupancy a beam leaves in a cell is then a continuous function of the beam's end point, i.e. of
  // cos / sin(pose heading + beam angle): only the reference's libm bits keep the map bit-equal)
  if (area) S("slam/occupancy_estimator/type", "area");
  S("slam/mapping/grid/type", "unbounded_plain");
  S("slam/map/height_in_meters", std::to_string(size_m));
  S("slam/map/width_in_meters", std::to_string(size_m));
  S("slam/map/meters_per_cell", "0.1");
  S("slam/scmtch/spe/type", "wmpp");
  if (matcher == 0) {
    S("slam/scmtch/type", "MC");
    S("slam/scmtch/MC/dispersion/translation", "0.2");
    S("slam/scmtch/MC/dispersion/rotation", "0.1");
    S("slam/scmtch/MC/dispersion/failed_attempts_limit", "20");
    S("slam/scmtch/MC/attempts_limit", "100");
    S("slam/scmtch/MC/seed", std::to_string(seed));
  } else {
    S("slam/scmtch/type", "HC");
    S("slam/scmtch/HC/distortion/translation", "0.1");
    S("slam/scmtch/HC/distortion/rotation", "0.1");
    S("slam/scmtch/HC/distortion/failed_attempts_limit", "6");
  }
  S("slam/scmtch/hip/strict", strict ? "true" : "false");
}

struct Calls : public GridScanMatcherObserver {
  long tests = 0, updates = 0;
  const char *tag = nullptr;  // REFWORLD_TRACE: print every event
  void on_scan_test(const RobotPose &p, const LaserScan2D &s, double score) override {
    ++tests;
    if (tag) fprintf(stderr, "%s test %.17g %.17g %.17g -> %.17g (%zu pts)\n", tag, p.x, p.y, p.theta, score, s.points().size());
  }
  void on_pose_update(const RobotPose &p, const LaserScan2D &, double score) override {
    ++updates;
    if (tag) fprintf(stderr, "%s update %.17g %.17g %.17g -> %.17g\n", tag, p.x, p.y, p.theta, score);
  }
};

}  // namespace

extern "C" {

// preset: 0 tinySLAM, 1 vinySLAM; matcher: 0 MC, 1 HC; wrap: 1 = map wrapped in HipMirroredGridMap
// (dirty log), 0 = bare reference map (full compare before every match)
// poses_out: n_scans x 6 (reference x, y, theta, HIP x, y, theta) after every scan
// out = {pose mismatches (bitwise), max |pose diff|, map cells compared, map cell mismatches,
//        max |occupancy diff|, ref scorer calls, hip scorer calls, ref accepted, hip accepted,
//        geometry equal, full uploads, re-binds, cells sent through the dirty path, final map width, height}
int refworld_compare(int preset, int matcher, int wrap, int n_scans, int n_beams, int strict, double size_m,
                     double *poses_out, double *out) {
  const double scale = 0.1;
  auto gt = std::make_shared<UnboundedPlainGridMap>(std::make_shared<MockGridCell>(0.0),
                                                    GridMapParams{300, 300, scale});
  {
    using C = CecumTextRasterMapPrimitive;
    C c1{81, 60, C::BoundPosition::Top}, c2{31, 21, C::BoundPosition::Bot};
    GridMapPatcher{}.apply_text_raster(*gt, c1.to_stream(), DiscretePoint2D{-40, 35}, 1, 1);
    GridMapPatcher{}.apply_text_raster(*gt, c2.to_stream(), DiscretePoint2D{-15, -6}, 1, 1);
  }
  MapPropertiesProvider props;
  fill_props(props, preset, matcher, 424242u, strict, size_m);
  auto ref = init_1h_slam(props);
  slamhip_ctx *ctx = nullptr;
  if (slamhip_ctx_create(0, &ctx) != SLAMHIP_OK) {
    std::cerr << "refworld: " << slamhip_last_error() << std::endl;
    return -1;
  }
  auto hip = init_hip_1h_slam(props, ctx, 0, wrap != 0);
  auto c_ref = std::make_shared<Calls>(), c_hip = std::make_shared<Calls>();
  ref->add_sm_observer(c_ref);
  hip->add_sm_observer(c_hip);

  // the robot drives up the corridor and turns a little; odometry carries a deterministic error that
  // the matcher has to take out again
  // (the `ahr` presets start a quarter of a radian off the walls' direction: with walls along the sensor frame's
  // x axis the segment between two neighbouring points has d_y ~ 1e-17 > 0 and d_x < 0, AngleHistogram's acos
  // returns pi exactly and the REFERENCE's own assert(angle < M_PI) ends the process, angle_histogram.h:90)
  RobotPose truth{scale / 2, scale / 2 - 6 * scale, deg2rad(90) + (preset >= 2 ? 0.25 : 0.0)};
  RobotPose prev_odom{0, 0, 0};
  long pose_mis = 0;
  double worst_pose = 0;
  for (int k = 0; k < n_scans; ++k) {
    TransformedLaserScan ts;
    ts.scan = LaserScanGenerator{to_lsp(15, 270, n_beams)}.laser_scan_2D(*gt, truth, 1);
    ts.quality = 1.0;
    const double ex = 0.03 * std::sin(1.7 * k), ey = -0.025 * std::cos(0.9 * k), et = 0.02 * std::sin(0.6 * k + 1);
    RobotPose odom{truth.x + ex, truth.y + ey, truth.theta + et};
    ts.pose_delta = k == 0 ? RobotPoseDelta{truth.x, truth.y, truth.theta}
                           : RobotPoseDelta{odom.x - prev_odom.x, odom.y - prev_odom.y, odom.theta - prev_odom.theta};
    prev_odom = k == 0 ? truth : odom;
    TransformedLaserScan ts_hip = ts;
    ref->handle_sensor_data(ts);
    hip->handle_sensor_data(ts_hip);
    const RobotPose pr = ref->pose(), ph = hip->pose();
    poses_out[6 * k + 0] = pr.x; poses_out[6 * k + 1] = pr.y; poses_out[6 * k + 2] = pr.theta;
    poses_out[6 * k + 3] = ph.x; poses_out[6 * k + 4] = ph.y; poses_out[6 * k + 5] = ph.theta;
    if (std::memcmp(&poses_out[6 * k], &poses_out[6 * k + 3], 3 * sizeof(double)) != 0) ++pose_mis;
    worst_pose = std::max({worst_pose, std::fabs(pr.x - ph.x), std::fabs(pr.y - ph.y), std::fabs(pr.theta - ph.theta)});
    // next true pose
    truth = RobotPose{truth.x + 0.04 * std::cos(0.35 * k), truth.y + 0.09, truth.theta + 0.015 * std::sin(0.8 * k)};
    // keep the generator's "not on a cell boundary" precondition
    if (std::fabs(truth.x / scale - std::round(truth.x / scale)) < 1e-3) truth.x += 0.013;
    if (std::fabs(truth.y / scale - std::round(truth.y / scale)) < 1e-3) truth.y += 0.013;
  }
  const GridMap &mr = ref->map(), &mh = hip->map();
  const bool geom = mr.width() == mh.width() && mr.height() == mh.height() && mr.origin() == mh.origin() &&
                    mr.scale() == mh.scale();
  long cells = 0, cell_mis = 0;
  double worst_occ = 0;
  if (geom) {
    const auto org = mr.origin();
    for (int y = 0; y < mr.height(); ++y)
      for (int x = 0; x < mr.width(); ++x) {
        const GridMap::Coord c{x - org.x, y - org.y};
        const Occupancy a = mr[c].occupancy(), b = mh[c].occupancy();
        ++cells;
        if (std::memcmp(&a.prob_occ, &b.prob_occ, sizeof(double)) != 0 ||
            std::memcmp(&a.estimation_quality, &b.estimation_quality, sizeof(double)) != 0) {
          ++cell_mis;
          worst_occ = std::max(worst_occ, std::fabs(a.prob_occ - b.prob_occ));
        }
      }
  }
  auto hgsm = std::dynamic_pointer_cast<HipGridScanMatcher>(hip->scan_matcher());
  out[0] = double(pose_mis);
  out[1] = worst_pose;
  out[2] = double(cells);
  out[3] = double(cell_mis);
  out[4] = worst_occ;
  out[5] = double(c_ref->tests);
  out[6] = double(c_hip->tests);
  out[7] = double(c_ref->updates);
  out[8] = double(c_hip->updates);
  out[9] = geom ? 1 : 0;
  out[10] = hgsm ? double(hgsm->mirror().full_uploads()) : -1;
  out[11] = hgsm ? double(hgsm->mirror().rebinds()) : -1;
  out[12] = hgsm ? double(hgsm->mirror().cells_sent()) : -1;
  out[13] = mr.width();
  out[14] = mr.height();
  hip.reset();
  hgsm.reset();
  slamhip_ctx_destroy(ctx);
  return 0;
}

// The same loop with the HIP world whose map is RESIDENT in HBM (host/slamhip_resident_world.h: match and map
// update both on the GPU, the window grows by itself, no host map): poses after every scan and the final
// map -- payload by payload: occupancy for tinySLAM's cells, the four belief masses for vinySLAM's -- against the
// reference world.
// out = {pose mismatches (bitwise), max |pose diff|, map cells compared, payload mismatches, max |payload diff|,
//        ref scorer calls, hip scorer calls, ref accepted, hip accepted, times the HBM window grew,
//        final reference width, height, final HBM window width, height, cell updates on the GPU, view mismatches,
//        seconds inside the reference world's handle_sensor_data (scans 1 .. n-1), wall seconds of the resident
//        world's loop over the same scans, run back to back and with its last queued update finished}  (18 doubles)
int refworld_compare_resident(int preset, int matcher, int n_scans, int n_beams, int strict, double size_m,
                              double *poses_out, double *out) {
  const double scale = 0.1;
  auto gt = std::make_shared<UnboundedPlainGridMap>(std::make_shared<MockGridCell>(0.0),
                                                    GridMapParams{300, 300, scale});
  {
    using C = CecumTextRasterMapPrimitive;
    C c1{81, 60, C::BoundPosition::Top}, c2{31, 21, C::BoundPosition::Bot};
    GridMapPatcher{}.apply_text_raster(*gt, c1.to_stream(), DiscretePoint2D{-40, 35}, 1, 1);
    GridMapPatcher{}.apply_text_raster(*gt, c2.to_stream(), DiscretePoint2D{-15, -6}, 1, 1);
  }
  MapPropertiesProvider props;
  fill_props(props, preset, matcher, 424242u, strict, size_m);
  auto ref = init_1h_slam(props);
  slamhip_ctx *ctx = nullptr;
  if (slamhip_ctx_create(0, &ctx) != SLAMHIP_OK) {
    std::cerr << "refworld: " << slamhip_last_error() << std::endl;
    return -1;
  }
  auto hip = init_hip_resident_1h_slam(props, ctx, 0);
  auto c_ref = std::make_shared<Calls>(), c_hip = std::make_shared<Calls>();
  ref->add_sm_observer(c_ref);
  // REFWORLD_NO_OBSERVER: nobody listens to the HIP world's matcher -- the way it runs in production -- and the
  // adapter filters the raw scan through slamhip_scan_filter_upload instead of the reference's filter_scan; the
  // scorer-call counters of the HIP side are then not available (reported equal to the reference's)
  const bool observe_hip = !std::getenv("REFWORLD_NO_OBSERVER");
  if (observe_hip) hip->add_sm_observer(c_hip);
  // (the `ahr` presets start a quarter of a radian off the walls' direction: with walls along the sensor frame's
  // x axis the segment between two neighbouring points has d_y ~ 1e-17 > 0 and d_x < 0, AngleHistogram's acos
  // returns pi exactly and the REFERENCE's own assert(angle < M_PI) ends the process, angle_histogram.h:90)
  RobotPose truth{scale / 2, scale / 2 - 6 * scale, deg2rad(90) + (preset >= 2 ? 0.25 : 0.0)};
  RobotPose prev_odom{0, 0, 0};
  long pose_mis = 0;
  double worst_pose = 0, ref_seconds = 0, hip_seconds = 0;
  // the scans first (they depend on the true trajectory only), then the two worlds ONE AFTER THE OTHER: interleaved
  // scan by scan -- as this loop ran until r03 -- the resident world's queued map update had the milliseconds of
  // the reference's next scan to finish in, and its cost never showed in the resident world's time
  std::vector<TransformedLaserScan> scans;
  for (int k = 0; k < n_scans; ++k) {
    TransformedLaserScan ts;
    ts.scan = LaserScanGenerator{to_lsp(15, 270, n_beams)}.laser_scan_2D(*gt, truth, 1);
    ts.quality = 1.0;
    const double ex = 0.03 * std::sin(1.7 * k), ey = -0.025 * std::cos(0.9 * k), et = 0.02 * std::sin(0.6 * k + 1);
    RobotPose odom{truth.x + ex, truth.y + ey, truth.theta + et};
    ts.pose_delta = k == 0 ? RobotPoseDelta{truth.x, truth.y, truth.theta}
                           : RobotPoseDelta{odom.x - prev_odom.x, odom.y - prev_odom.y, odom.theta - prev_odom.theta};
    prev_odom = k == 0 ? truth : odom;
    scans.push_back(ts);
    truth = RobotPose{truth.x + 0.04 * std::cos(0.35 * k), truth.y + 0.09, truth.theta + 0.015 * std::sin(0.8 * k)};
    if (std::fabs(truth.x / scale - std::round(truth.x / scale)) < 1e-3) truth.x += 0.013;
    if (std::fabs(truth.y / scale - std::round(truth.y / scale)) < 1e-3) truth.y += 0.013;
  }
  auto tag = [&](int k) {
    if (!std::getenv("REFWORLD_TRACE")) return;
    const bool on = k == std::atoi(std::getenv("REFWORLD_TRACE"));
    c_ref->tag = on ? "ref" : nullptr;
    c_hip->tag = on ? "hip" : nullptr;
  };
  for (int k = 0; k < n_scans; ++k) {
    TransformedLaserScan ts = scans[k];
    tag(k);
    const auto t0 = std::chrono::steady_clock::now();
    ref->handle_sensor_data(ts);
    const auto t1 = std::chrono::steady_clock::now();
    if (k > 0) ref_seconds += std::chrono::duration<double>(t1 - t0).count();  // (the first scan pays the allocations)
    const RobotPose pr = ref->pose();
    poses_out[6 * k + 0] = pr.x; poses_out[6 * k + 1] = pr.y; poses_out[6 * k + 2] = pr.theta;
  }
  std::chrono::steady_clock::time_point hip_t0;
  for (int k = 0; k < n_scans; ++k) {
    TransformedLaserScan ts = scans[k];
    tag(k);
    if (k == 1) hip_t0 = std::chrono::steady_clock::now();  // (scan 0 done: allocations, first bind)
    hip->handle_sensor_data(ts);
    const RobotPose ph = hip->pose();
    poses_out[6 * k + 3] = ph.x; poses_out[6 * k + 4] = ph.y; poses_out[6 * k + 5] = ph.theta;
  }
  slamhip_ctx_synchronize(ctx);  // the last scan's queued map update belongs to the loop
  hip_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - hip_t0).count();
  for (int k = 0; k < n_scans; ++k) {
    const double *pr = &poses_out[6 * k], *ph = &poses_out[6 * k + 3];
    if (std::memcmp(pr, ph, 3 * sizeof(double)) != 0) ++pose_mis;
    worst_pose = std::max({worst_pose, std::fabs(pr[0] - ph[0]), std::fabs(pr[1] - ph[1]), std::fabs(pr[2] - ph[2])});
  }
  // the final maps, payload by payload over the reference's extent (the HBM window is a superset or the cells
  // outside it were never touched: they must hold the prototype's payload in the reference map)
  const GridMap &mr = ref->map();
  int model = 0, w = 0, h = 0, ox = 0, oy = 0;
  long long grown = 0;
  slamhip_map_info(ctx, 0, &model, &w, &h, &ox, &oy, nullptr, &grown);
  const int st = model == SLAMHIP_CELL_TBM ? 4 : 1;
  std::vector<double> dev((size_t)w * h * st);
  if (slamhip_map_download_window(ctx, 0, 0, 0, w, h, dev.data()) != SLAMHIP_OK) {
    std::cerr << "refworld: " << slamhip_last_error() << std::endl;
    return -1;
  }
  auto payload = [&](const GridCell &c, double *p) {
    if (st == 4) {
      const auto &b = static_cast<const TbmBaseCell &>(c).belief();
      p[0] = b.unknown(); p[1] = b.empty(); p[2] = b.occupied(); p[3] = b.conflict();
    } else {
      p[0] = c.occupancy().prob_occ;
    }
  };
  double unk[4] = {0, 0, 0, 0};
  payload(*mr.new_cell(), unk);
  long cells = 0, cell_mis = 0;
  double worst = 0;
  const auto org = mr.origin();
  for (int y = 0; y < mr.height(); ++y)
    for (int x = 0; x < mr.width(); ++x) {
      const GridMap::Coord c{x - org.x, y - org.y};
      double a[4], b[4];
      payload(mr[c], a);
      const int ix = c.x + ox, iy = c.y + oy;
      if (0 <= ix && ix < w && 0 <= iy && iy < h) std::memcpy(b, &dev[((size_t)iy * w + ix) * st], st * sizeof(double));
      else std::memcpy(b, unk, st * sizeof(double));
      ++cells;
      if (std::memcmp(a, b, st * sizeof(double)) != 0) {
        ++cell_mis;
        for (int k = 0; k < st; ++k) worst = std::max(worst, std::fabs(a[k] - b[k]));
      }
    }
  // and the view a map consumer gets: occupancy of a row of cells through GridMap::operator[]
  long view_mis = 0;
  for (int x = 0; x < mr.width(); ++x) {
    const GridMap::Coord c{x - org.x, 0};
    const Occupancy a = mr[c].occupancy(), b = hip->map()[c].occupancy();
    if (std::memcmp(&a.prob_occ, &b.prob_occ, sizeof(double)) != 0) ++view_mis;
  }
  out[0] = double(pose_mis);
  out[1] = worst_pose;
  out[2] = double(cells);
  out[3] = double(cell_mis);
  out[4] = worst;
  out[5] = double(c_ref->tests);
  out[6] = double(observe_hip ? c_hip->tests : c_ref->tests);
  out[7] = double(c_ref->updates);
  out[8] = double(observe_hip ? c_hip->updates : c_ref->updates);
  out[9] = double(grown);
  out[10] = mr.width();
  out[11] = mr.height();
  out[12] = w;
  out[13] = h;
  out[14] = double(hip->cell_updates());
  out[15] = double(view_mis);
  out[16] = ref_seconds;  // wall time inside handle_sensor_data, scans 1 .. n-1: the reference's world ...
  out[17] = hip_seconds;  // ... and the resident world (GPU match + GPU map update), same scans, same process
  hip.reset();
  slamhip_ctx_destroy(ctx);
  return 0;
}

}  // extern "C"
