// slamhip_resident_world.h -- the single-hypothesis world (tinySLAM / vinySLAM) with its map RESIDENT in HBM.
//
// init_hip_1h_slam (slamhip_init_slam.h) keeps the reference's world, map and scan adder and mirrors the host
// map into HBM before every match.  Here the map never exists on the host: one scan is
//     match on the GPU (slamhip_matcher_process_scan)  ->  update of the same HBM window (slamhip_map_append_scan, K6)
// which is SingleStateHypothesisLaserScanGridWorld::handle_observation
// (src/core/states/single_state_hypothesis_laser_scan_grid_world.h:52-65) with GridMapScanAdder::append_scan
// (src/core/maps/grid_map_scan_adders.h:54-75) replaced by the device's.  The window grows like the reference's
// unbounded map (slamhip_map_set_auto_grow).  map() hands out a read-only view that fetches 64x64-cell chunks
// from the GPU when a consumer (a map dumper, the ROS occupancy-grid publisher) looks at them.
// Compiled only with the reference headers on the include path; contains no reference code.
// oracle/ref_world_harness.cpp runs it next to init_1h_slam.
#ifndef SLAMHIP_RESIDENT_WORLD_H
#define SLAMHIP_RESIDENT_WORLD_H

#include <cstring>
#include <functional>
#include <limits>
#include <memory>
#include <unordered_map>
#include <vector>

#include "core/states/laser_scan_grid_world.h"
#include "slamhip_init_scan_matching.h"

class HipResidentWorld : public LaserScanGridWorld {
public:
  struct Config {
    double localized_scan_quality = 1.0, raw_scan_quality = 1.0;  // SingleStateHypothesisLSGWProperties
    std::size_t scan_margin = 0;
    slamhip_scan_adder_cfg adder{};  // init_scan_adder + the cell kind of init_occupied_area_model
    int cell_model = SLAMHIP_CELL_OCC;
    GridMapParams map{100, 100, 0.1};
    double unknown[4] = {0.5, 0, 0, 0};  // the cell prototype's payload
    int map_id = 0;
    int tbm_kind = 0;  // TBM cells: 0 tbm_consistent, 1 tbm_unknown_even_occ (what map() reports as occupancy)
    int omqe = 0;      // observation quality estimator (init_omqe): 0 idle, 1 ahr (AngleHistogramResiprocalOMQE)
    HipScanTrig trig{};
    bool raw_exact = false;  // slam/scmtch/hip/strict with the raw trig provider: see handle_observation
  };
  HipResidentWorld(slamhip_ctx *ctx, std::shared_ptr<HipGridScanMatcher> gsm, const Config &cfg)
      : _ctx{ctx}, _gsm{std::move(gsm)}, _cfg{cfg} {
    // RegularSquaresGrid starts with the origin in the middle (regular_squares_grid.h:120-122)
    const int w = cfg.map.width_cells, h = cfg.map.height_cells;
    slamhip_or_die(slamhip_map_bind(ctx, cfg.map_id, cfg.cell_model, w, h, w / 2, h / 2, cfg.map.meters_per_cell,
                                    cfg.unknown), "map_bind");
    slamhip_or_die(slamhip_map_set_auto_grow(ctx, cfg.map_id, 1), "map_set_auto_grow");
    // the pose of a scan is handed on while the GPU still writes that scan into the map: the next match, and every
    // read of the map, is ordered behind the update on the context's stream
    slamhip_or_die(slamhip_map_set_deferred(ctx, 1), "map_set_deferred");
    _view = std::make_shared<HipResidentMapView>(ctx, cfg.map_id, cfg.map, 0.5, cfg.tbm_kind);
  }

  auto scan_matcher() { return std::static_pointer_cast<GridScanMatcher>(_gsm); }
  void add_sm_observer(std::shared_ptr<GridScanMatcherObserver> obs) { _gsm->subscribe(obs); }
  void remove_sm_observer(std::shared_ptr<GridScanMatcherObserver> obs) { _gsm->unsubscribe(obs); }
  const GridMap &map() const override { return *_view; }
  using LaserScanGridWorld::map;
  long long cell_updates() {  // (waits for the updates that are still queued)
    long long nu = 0;
    slamhip_or_die(slamhip_map_drain(_ctx, &nu), "map_drain");
    _cell_updates += nu;
    return _cell_updates;
  }

  void handle_observation(TransformedLaserScan &tr_scan) override {
    _gsm->reset_state();
    auto pose_delta = RobotPoseDelta{};
    _gsm->process_scan(tr_scan, pose(), map(), pose_delta);  // the mirror is resident: nothing is uploaded
    update_robot_pose(pose_delta);
    tr_scan.quality = pose_delta ? _cfg.localized_scan_quality : _cfg.raw_scan_quality;

    // GridMapScanAdder::append_scan (grid_map_scan_adders.h:54-75): the points [margin, n - margin - 1]; the
    // observation quality estimator is reset on the WHOLE scan and asked per point index (:63-71)
    const auto &pts = tr_scan.scan.points();
    if (pts.empty()) return;
    const size_t first = _cfg.scan_margin, last = pts.size() - _cfg.scan_margin - 1;
    std::vector<double> &r = _r, &a = _a, &q = _q;
    std::vector<int> &occ = _occ;
    r.clear();
    a.clear();
    occ.clear();
    q.clear();
    if (_cfg.omqe) {
      _all_r.resize(pts.size());
      _all_a.resize(pts.size());
      _all_q.resize(pts.size());
      for (size_t i = 0; i < pts.size(); ++i) {
        _all_r[i] = pts[i].range();
        _all_a[i] = pts[i].angle();
      }
      slamhip_or_die(slamhip_omqe_quality(_cfg.omqe, (int)pts.size(), _all_r.data(), _all_a.data(), _all_q.data()),
                     "omqe_quality");
    }
    for (size_t i = first; i <= last && i < pts.size(); ++i) {
      r.push_back(pts[i].range());
      a.push_back(pts[i].angle());
      occ.push_back(pts[i].is_occupied() ? 1 : 0);
      if (_cfg.omqe) q.push_back(_all_q[i]);
    }
    const int n = (int)r.size();
    std::vector<double> &c = _c, &s = _s;
    if (a != _trig_a) {  // (a scanner's angles do not change from scan to scan: their cos / sin are made once)
      c.resize(n);
      s.resize(n);
      if (_cfg.trig.mode == SLAMHIP_TRIG_CACHED)
        slamhip_or_die(slamhip_beam_trig_cached(n, a.data(), _cfg.trig.a_min, _cfg.trig.a_max, _cfg.trig.a_inc, c.data(),
                                                s.data()), "beam_trig_cached");
      else
        slamhip_or_die(slamhip_beam_trig_raw(n, a.data(), c.data(), s.data()), "beam_trig_raw");
      _trig_a = a;
    }
    slamhip_scan_adder_cfg adder = _cfg.adder;
    adder.scan_quality = tr_scan.quality;
    const RobotPose p = pose();
    const double p3[3] = {p.x, p.y, p.theta};
    long long nu = 0;
    // strict with the raw provider (the reference's default): the scan adder's end points are cos / sin(theta + a) by the
    // host's libm, the reference's bits -- what the area estimator's occupancies and TBM / GMapping payloads depend on
    // continuously (slamhip_map_append_scan_raw); else the provider's table and the cached provider's angle addition
    if (_cfg.raw_exact && _cfg.trig.mode != SLAMHIP_TRIG_CACHED)
      slamhip_or_die(slamhip_map_append_scan_raw(_ctx, _cfg.map_id, &adder, p3, n, r.data(), a.data(), occ.data(),
                                                 _cfg.omqe ? q.data() : nullptr, &nu), "map_append_scan_raw");
    else
      slamhip_or_die(slamhip_map_append_scan_q(_ctx, _cfg.map_id, &adder, p3, n, r.data(), c.data(), s.data(), occ.data(),
                                               _cfg.omqe ? q.data() : nullptr, &nu), "map_append_scan");
    if (nu > 0) _cell_updates += nu;  // (-1: queued, counted by cell_updates())
    _view->invalidate();
  }

private:
  slamhip_ctx *_ctx;
  std::shared_ptr<HipGridScanMatcher> _gsm;
  Config _cfg;
  std::shared_ptr<HipResidentMapView> _view;
  long long _cell_updates = 0;
  std::vector<double> _r, _a, _q, _c, _s, _all_r, _all_a, _all_q, _trig_a;  // per-scan buffers, kept between scans
  std::vector<int> _occ;
};

// the factory next to init_1h_slam (src/utils/init_slam.h:12-25): same properties
inline std::shared_ptr<HipResidentWorld> init_hip_resident_1h_slam(const PropertiesProvider &props,
                                                                   slamhip_ctx *ctx = nullptr, int map_id = 0) {
  if (!ctx) slamhip_or_die(slamhip_ctx_create(props.get_int("slam/scmtch/hip/device", 0), &ctx), "ctx_create");
  HipResidentWorld::Config cfg;
  std::tie(cfg.localized_scan_quality, cfg.raw_scan_quality) = init_pose_quality_estimators(props);
  cfg.map = init_grid_map_params(props);
  cfg.map_id = map_id;
  const auto grid = props.get_str("slam/mapping/grid/type", "<undefined>");
  if (grid != "unbounded_plain" && grid != "unbounded_lazy_tiled") {
    std::cerr << "the resident world keeps an unbounded dense window; grid type " << grid << " is outside it" << std::endl;
    std::exit(-1);
  }
  // init_occupied_area_model (init_occupancy_mapping.h:94-113): the cell kind is the update rule
  const auto area = props.get_str("slam/mapping/grid/area/type", "<undefined>");
  const bool occupancy_oie = props.get_str(Slam_SM_NS + "oie/type", "discrepancy") == "occupancy";
  if (area.rfind("tbm", 0) == 0) {
    if (occupancy_oie) {
      std::cerr << "TBM cells scored through the occupancy OIE need the host map (init_hip_1h_slam)" << std::endl;
      std::exit(-1);
    }
    cfg.cell_model = SLAMHIP_CELL_TBM;
    cfg.tbm_kind = area == "tbm_unknown_even_occ" ? 1 : 0;
    cfg.adder.rule = SLAMHIP_RULE_TBM;
    const double vacuous[4] = {1.0, 0.0, 0.0, 0.0};  // TbmBaseCell: total ignorance (tbm_grid_cells.h:12-19)
    std::memcpy(cfg.unknown, vacuous, sizeof(vacuous));
  } else if (area == "mean_probability" || area == "affine_quality_merge") {
    cfg.cell_model = SLAMHIP_CELL_OCC;
    cfg.adder.rule = area == "mean_probability" ? SLAMHIP_RULE_MEAN : SLAMHIP_RULE_AFFINE;
    cfg.unknown[0] = 0.5;  // Occupancy{0.5, 1} (naive_grid_cells.h:10,29)
  } else {
    std::cerr << "Unknown occupied area type: " << area << std::endl;
    std::exit(-1);
  }
  // init_occ_estimator / init_scan_adder (init_occupancy_mapping.h:38-92)
  cfg.adder.base_occupied_prob = props.get_dbl("slam/occupancy_estimator/base_occupied/prob", 0.95);
  cfg.adder.base_occupied_qual = props.get_dbl("slam/occupancy_estimator/base_occupied/qual", 1.0);
  cfg.adder.base_empty_prob = props.get_dbl("slam/occupancy_estimator/base_empty/prob", 0.01);
  cfg.adder.base_empty_qual = props.get_dbl("slam/occupancy_estimator/base_empty/qual", 1.0);
  const auto est = props.get_str("slam/occupancy_estimator/type", "const");
  if (est != "const" && est != "area") {
    std::cerr << "Unknown estimator type: " << est << std::endl;
    std::exit(-1);
  }
  cfg.adder.occupancy_estimator = est == "area" ? 1 : 0;
  // init_omqe (init_occupancy_mapping.h:64-80; the key really ends in "typetype")
  const auto omqe = props.get_str("slam/mapping/observation_quality_estimator/typetype", "idle");
  if (omqe != "idle" && omqe != "ahr") {
    std::cerr << "[ERROR] Unknown OMQE type: " << omqe << std::endl;
    std::exit(-1);
  }
  cfg.omqe = omqe == "ahr" ? 1 : 0;
  cfg.adder.blur = props.get_dbl("slam/mapping/blur", 0.0);
  cfg.adder.max_range = props.get_dbl("slam/mapping/max_range", std::numeric_limits<double>::infinity());
  cfg.adder.scan_quality = 1.0;
  cfg.raw_exact = props.get_bool(Slam_SM_NS + "hip/strict", false) && !props.get_bool(Slam_SM_NS + "hip/trig_cache", false);
  auto gsm = std::dynamic_pointer_cast<HipGridScanMatcher>(init_hip_scan_matcher(props, ctx, map_id));
  gsm->set_resident_map(true);
  return std::make_shared<HipResidentWorld>(ctx, gsm, cfg);
}

#endif  // SLAMHIP_RESIDENT_WORLD_H
