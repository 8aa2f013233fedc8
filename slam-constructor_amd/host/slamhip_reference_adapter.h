// slamhip_reference_adapter.h -- the reference-side binding of the C-ABI (include/slamhip.h).
//
// This header is what a slam-constructor maintainer adds to the reference tree (see
// INTEGRATION.md).  It is compiled ONLY with the reference headers on the include path
// (-I<reference>/src) and is not part of libslamhip.so.  It contains no reference code: it
// derives from the reference's own plugin interfaces
//   GridScanMatcher              src/core/scan_matchers/grid_scan_matcher.h:138-227
//   ScanProbabilityEstimator     src/core/scan_matchers/grid_scan_matcher.h:85-136
//   GridMap                      src/core/maps/grid_map.h:22-74
// and forwards the hot path to the GPU library.
//
//   HipMirroredGridMap     GridMap decorator: forwards everything to the wrapped map and logs the
//                          cells touched by update()/reset() (same idea as
//                          RescalableCachingGridMap::update, rescalable_caching_grid_map.h:100-105)
//   HipMapMirror           keeps the dense HBM window of one GridMap in sync (full upload on first
//                          use / growth, dirty log afterwards)
//   HipGridScanMatcher     GridScanMatcher whose process_scan runs the MC / HC / BF accept chain
//                          through slamhip_matcher_process_scan; observers are fed from the replay
//   HipScanProbabilityEstimator  ScanProbabilityEstimator whose estimate_scan_probability is one
//                          slamhip_score_poses call (used by code that scores single poses)
//
// Error convention of the reference (no exceptions; bad config -> message + std::exit(-1),
// init_scan_matching.h:39-43): a failing slamhip call prints slamhip_last_error() and exits.
#ifndef SLAMHIP_REFERENCE_ADAPTER_H
#define SLAMHIP_REFERENCE_ADAPTER_H

#include <cstdlib>
#include <cstring>
#include <iostream>
#include <limits>
#include <memory>
#include <typeinfo>
#include <utility>
#include <unordered_map>
#include <vector>

#include "core/maps/grid_map.h"
#include "core/maps/tbm_grid_cells.h"
#include "core/scan_matchers/grid_scan_matcher.h"
#include "core/scan_matchers/weighted_mean_point_probability_spe.h"
#include "slamhip.h"

#ifndef SLAMHIP_GMAPPING_OBSTACLE
// GmappingBaseCell keeps its mean obstacle point private and has no accessor (gmapping_grid_cell.h:40-42).  The
// reference stays unpatched: the arguments of an explicit template instantiation may name a private member (access
// checks do not apply to them, [temp.spec]), and the instantiation hands the member pointer to a friend function.
// (A build that would rather add `const Point2D &obstacle() const { return obst; }` to the cell defines this macro
// as `static_cast<const GmappingBaseCell &>(cell).obstacle()` before including this header.)
#if __has_include("slams/gmapping/gmapping_grid_cell.h")
#include "slams/gmapping/gmapping_grid_cell.h"
namespace {  // (one instantiation per translation unit)
template <typename Tag, typename Tag::type Member>
struct SlamhipPrivateMember {
  friend typename Tag::type slamhip_private_member(Tag) { return Member; }
};
struct SlamhipGmappingObst {
  using type = Point2D GmappingBaseCell::*;
  friend type slamhip_private_member(SlamhipGmappingObst);
};
template struct SlamhipPrivateMember<SlamhipGmappingObst, &GmappingBaseCell::obst>;
inline const Point2D &slamhip_gmapping_obstacle(const GridCell &cell) {
  return static_cast<const GmappingBaseCell &>(cell).*slamhip_private_member(SlamhipGmappingObst{});
}
}  // namespace
#define SLAMHIP_GMAPPING_OBSTACLE(cell) slamhip_gmapping_obstacle(cell)
#else
#define SLAMHIP_GMAPPING_OBSTACLE(cell) \
  (slamhip_or_die(SLAMHIP_ERR_UNSUPPORTED, "GMapping cells: slams/gmapping/gmapping_grid_cell.h is not on the include path"), Point2D{0, 0})
#endif
#endif

inline void slamhip_or_die(int rc, const char *what) {
  if (rc == SLAMHIP_OK) return;
  std::cerr << "[slamhip] " << what << ": " << slamhip_last_error() << std::endl;
  std::exit(-1);
}

// ------------------------------------------------------------------------------------------------
// GridMap decorator that remembers which cells update()/reset() touched since the mirror last looked.
// Every cell is logged once per interval (a scan touches the robot's own cell once per beam).
class HipMirroredGridMap : public GridMap {
public:
  explicit HipMirroredGridMap(std::shared_ptr<GridMap> wrapped)
      : GridMap{std::shared_ptr<GridCell>(wrapped->new_cell().release()),
                GridMapParams{wrapped->width(), wrapped->height(), wrapped->scale()}},
        _map{wrapped} {}
  HipMirroredGridMap(std::shared_ptr<GridMap> wrapped, std::shared_ptr<GridCell> prototype)
      : GridMap{prototype, GridMapParams{wrapped->width(), wrapped->height(), wrapped->scale()}},
        _map{wrapped} {}

  void update(const Coord &c, const AreaOccupancyObservation &aoo) override {
    _map->update(c, aoo);
    log(c);
  }
  void reset(const Coord &c, const GridCell &cell) override {
    _map->reset(c, cell);
    log(c);
  }
  const GridCell &operator[](const Coord &c) const override { return (*_map)[c]; }
  int width() const override { return _map->width(); }
  int height() const override { return _map->height(); }
  double scale() const override { return _map->scale(); }
  void rescale(double s) override { _map->rescale(s); }
  DiscretePoint2D origin() const override { return _map->origin(); }
  bool has_cell(const Coord &c) const override { return _map->has_cell(c); }
  bool validate() const override { return _map->validate(); }
  std::vector<char> save_state() const override { return _map->save_state(); }
  void load_state(const std::vector<char> &d) override {
    _map->load_state(d);
    _everything = true;
  }
  const std::shared_ptr<GridMap> &wrapped() const { return _map; }

  // the cells (external coordinates) touched since the last call; *everything = the whole map must
  // be taken anew (load_state)
  std::vector<Coord> take_dirty(bool *everything = nullptr) const {
    if (everything) *everything = _everything;
    _everything = false;
    ++_epoch;
    return std::exchange(_dirty, {});
  }

private:
  void log(const Coord &c) {
    // the stamp grid follows the wrapped map's geometry (an unbounded map grows inside update())
    const auto org = _map->origin();
    const int w = _map->width(), h = _map->height();
    if (w != _sw || h != _sh || org.x != _sox || org.y != _soy) {
      std::vector<unsigned> ns(size_t(w) * h, 0u);
      for (int y = 0; y < _sh; ++y) {
        const int ny = y - _soy + org.y;
        if (ny < 0 || ny >= h) continue;
        for (int x = 0; x < _sw; ++x) {
          const int nx = x - _sox + org.x;
          if (0 <= nx && nx < w) ns[size_t(ny) * w + nx] = _stamp[size_t(y) * _sw + x];
        }
      }
      _stamp.swap(ns);
      _sw = w; _sh = h; _sox = org.x; _soy = org.y;
    }
    const int ix = c.x + org.x, iy = c.y + org.y;
    if (ix < 0 || ix >= w || iy < 0 || iy >= h) {  // a bounded map ignores nothing: log it plainly
      _dirty.push_back(c);
      return;
    }
    unsigned &st = _stamp[size_t(iy) * w + ix];
    if (st == _epoch) return;
    st = _epoch;
    _dirty.push_back(c);
  }
  std::shared_ptr<GridMap> _map;
  mutable std::vector<Coord> _dirty;
  mutable unsigned _epoch = 1;
  mutable bool _everything = false;
  std::vector<unsigned> _stamp;
  int _sw = 0, _sh = 0, _sox = 0, _soy = 0;
};

// ------------------------------------------------------------------------------------------------
// Keeps the dense HBM window of one GridMap equal to the host map.  Correct for ANY GridMap: with a
// HipMirroredGridMap the cells its log names are re-sent; without one the whole map is compared with
// a host shadow of what the GPU holds and the cells that differ are re-sent (one virtual call per
// cell and scan -- wrap the map to avoid it).  Growth of an unbounded map is a re-bind (the device
// moves the old window by the origin shift, plain_grid_map.h:133-173), not a re-upload.
class HipMapMirror {
public:
  HipMapMirror(slamhip_ctx *ctx, int map_id, int cell_model, bool bounded)
      : _ctx{ctx}, _id{map_id}, _model{cell_model}, _bounded{bounded} {}

  int id() const { return _id; }
  bool bounded() const { return _bounded; }
  // counters for tests / logs: full uploads, re-binds on growth, cells sent through the dirty path
  long full_uploads() const { return _n_full; }
  long rebinds() const { return _n_rebind; }
  long cells_sent() const { return _n_cells; }

  // the window IS the map (host/slamhip_resident_world.h: matched and updated in HBM): nothing to mirror
  void set_resident(bool on) { _resident = on; }

  void sync(const GridMap &map, const HipMirroredGridMap *dirty_source = nullptr) {
    if (_resident) return;
    if (!dirty_source) dirty_source = dynamic_cast<const HipMirroredGridMap *>(&map);
    const auto org = map.origin();
    const bool same = _w == map.width() && _h == map.height() && _ox == org.x && _oy == org.y &&
                      _scale == map.scale();
    const int st = stride();
    bool everything = false;
    std::vector<GridMap::Coord> dirty;
    if (dirty_source) dirty = dirty_source->take_dirty(&everything);
    if (!same) {
      double unk[4] = {0, 0, 0, 0};
      payload(*map.new_cell(), unk);
      const bool grown = _w > 0 && _scale == map.scale() && !everything;
      slamhip_or_die(slamhip_map_bind(_ctx, _id, _model, map.width(), map.height(), org.x, org.y,
                                      map.scale(), unk), "map_bind");
      if (!dirty_source) reshape_shadow(map, org, grown, unk);
      _w = map.width(); _h = map.height(); _ox = org.x; _oy = org.y; _scale = map.scale();
      if (!grown) {
        upload_all(map, org, !dirty_source);
        return;
      }
      ++_n_rebind;
    } else if (everything) {
      upload_all(map, org, !dirty_source);
      return;
    }
    if (dirty_source) {
      _shadow.clear();
      send(map, org, dirty);
      return;
    }
    if (_shadow.size() != size_t(_w) * _h * st) {  // the map used to come with a log
      upload_all(map, org, true);
      return;
    }
    // no log: find the cells that differ from what the GPU holds
    std::vector<int> xy;
    std::vector<double> vals;
    double p[4];
    for (int y = 0; y < _h; ++y)
      for (int x = 0; x < _w; ++x) {
        payload(map[{x - org.x, y - org.y}], p);
        double *sh = &_shadow[(size_t(y) * _w + x) * st];
        if (std::memcmp(sh, p, st * sizeof(double)) == 0) continue;
        std::memcpy(sh, p, st * sizeof(double));
        xy.push_back(x);
        xy.push_back(y);
        vals.insert(vals.end(), p, p + st);
      }
    if (xy.empty()) return;
    _n_cells += long(xy.size() / 2);
    slamhip_or_die(slamhip_map_apply_dirty(_ctx, _id, int(xy.size() / 2), xy.data(), vals.data()),
                   "map_apply_dirty");
  }

private:
  int stride() const { return _model == SLAMHIP_CELL_TBM ? 4 : (_model == SLAMHIP_CELL_GMAPPING ? 3 : 1); }
  void payload(const GridCell &c, double *out) const {
    if (_model == SLAMHIP_CELL_TBM) {
      const auto &b = static_cast<const TbmBaseCell &>(c).belief();
      out[0] = b.unknown(); out[1] = b.empty(); out[2] = b.occupied(); out[3] = b.conflict();
    } else if (_model == SLAMHIP_CELL_GMAPPING) {
      const Point2D o = SLAMHIP_GMAPPING_OBSTACLE(c);
      out[0] = c.occupancy().prob_occ; out[1] = o.x; out[2] = o.y;
    } else {
      out[0] = c.occupancy().prob_occ;
    }
  }
  void upload_all(const GridMap &map, const DiscretePoint2D &org, bool keep_shadow) {
    const int st = stride();
    std::vector<double> buf(size_t(_w) * _h * st);
    for (int y = 0; y < _h; ++y)
      for (int x = 0; x < _w; ++x) payload(map[{x - org.x, y - org.y}], &buf[(size_t(y) * _w + x) * st]);
    slamhip_or_die(slamhip_map_upload_window(_ctx, _id, 0, 0, _w, _h, buf.data()), "map_upload_window");
    if (keep_shadow) _shadow.swap(buf);
    else _shadow.clear();
    ++_n_full;
  }
  void send(const GridMap &map, const DiscretePoint2D &org, const std::vector<GridMap::Coord> &dirty) {
    if (dirty.empty()) return;
    const int st = stride();
    std::vector<int> xy;
    std::vector<double> vals;
    xy.reserve(dirty.size() * 2);
    vals.reserve(dirty.size() * st);
    for (const auto &c : dirty) {
      xy.push_back(c.x + org.x);
      xy.push_back(c.y + org.y);
      double p[4];
      payload(map[c], p);
      vals.insert(vals.end(), p, p + st);
    }
    _n_cells += long(dirty.size());
    slamhip_or_die(slamhip_map_apply_dirty(_ctx, _id, int(dirty.size()), xy.data(), vals.data()),
                   "map_apply_dirty");
  }
  // the shadow follows a re-bind the way the device window does: old cells keep their external place,
  // new area holds the prototype payload
  void reshape_shadow(const GridMap &map, const DiscretePoint2D &org, bool grown, const double *unk) {
    const int st = stride(), nw = map.width(), nh = map.height();
    std::vector<double> ns(size_t(nw) * nh * st);
    for (size_t i = 0; i < size_t(nw) * nh; ++i) std::memcpy(&ns[i * st], unk, st * sizeof(double));
    if (grown && !_shadow.empty()) {
      const int dx = org.x - _ox, dy = org.y - _oy;
      for (int y = 0; y < _h; ++y) {
        const int ny = y + dy;
        if (ny < 0 || ny >= nh) continue;
        for (int x = 0; x < _w; ++x) {
          const int nx = x + dx;
          if (0 <= nx && nx < nw)
            std::memcpy(&ns[(size_t(ny) * nw + nx) * st], &_shadow[(size_t(y) * _w + x) * st], st * sizeof(double));
        }
      }
    }
    _shadow.swap(ns);
  }
  slamhip_ctx *_ctx;
  int _id, _model;
  bool _bounded;
  bool _resident = false;
  int _w = -1, _h = -1, _ox = 0, _oy = 0;
  double _scale = 0;
  std::vector<double> _shadow;  // only kept for maps without a dirty log
  long _n_full = 0, _n_rebind = 0, _n_cells = 0;
};

// ------------------------------------------------------------------------------------------------
// Uploads the filtered scan the way the scorer iterates it.  `weighting`: 0 even, 1 viny, 2 ahr
// (init_swp, src/utils/init_scan_matching.h:74-92).  Only RawTrigonometryProvider scans carry no
// table; for CachedTrigonometryProvider pass its update() arguments.
struct HipScanTrig {
  int mode = SLAMHIP_TRIG_RAW;
  double a_min = 0, a_max = 0, a_inc = 1;
};

inline void hip_upload_filtered_scan(slamhip_ctx *ctx, const LaserScan2D &scan, int weighting,
                                     const HipScanTrig &trig) {
  const auto &pts = scan.points();
  const int n = int(pts.size());
  std::vector<double> r(n), a(n), f(n), w(n), c(n), s(n);
  for (int i = 0; i < n; ++i) {
    r[i] = pts[i].range();
    a[i] = pts[i].angle();
    f[i] = pts[i].factor();
  }
  slamhip_or_die(slamhip_scan_weights(weighting, n, r.data(), a.data(), w.data()), "scan_weights");
  if (trig.mode == SLAMHIP_TRIG_CACHED)
    slamhip_or_die(slamhip_beam_trig_cached(n, a.data(), trig.a_min, trig.a_max, trig.a_inc, c.data(),
                                            s.data()), "beam_trig_cached");
  else
    slamhip_or_die(slamhip_beam_trig_raw(n, a.data(), c.data(), s.data()), "beam_trig_raw");
  slamhip_or_die(slamhip_scan_upload(ctx, n, r.data(), c.data(), s.data(), w.data(), f.data()),
                 "scan_upload");
  // (what SLAMHIP_POSE_TRIG_RAW_EXACT -- RawTrigonometryProvider bit for bit -- adds the pose heading to; a host-side
  // copy, sent to the device only if a scoring call asks for that mode)
  slamhip_or_die(slamhip_scan_set_angles(ctx, n, a.data()), "scan_set_angles");
}

// ------------------------------------------------------------------------------------------------
class HipGridScanMatcher : public GridScanMatcher {
public:
  // `spe` is the reference estimator the SLAM was configured with: it still does filter_scan on
  // the host (once per scan); scoring goes to the GPU.  `matcher` was created with
  // slamhip_matcher_create_{mc,hc,bf} using the same parameters init_scan_matcher would pass.
  HipGridScanMatcher(SPE spe, slamhip_ctx *ctx, slamhip_matcher *matcher,
                     std::shared_ptr<HipMapMirror> mirror, int weighting, HipScanTrig trig = {})
      : GridScanMatcher{spe}, _ctx{ctx}, _m{matcher}, _mirror{mirror}, _weighting{weighting},
        _trig{trig} {}
  ~HipGridScanMatcher() override { slamhip_matcher_destroy(_m); }

  void reset_state() override { slamhip_or_die(slamhip_matcher_reset_state(_m), "reset_state"); }
  // optional: a log kept somewhere else than in the map handed to process_scan (a map that IS a
  // HipMirroredGridMap is recognised by itself; any other map is compared cell by cell)
  void set_dirty_source(const HipMirroredGridMap *src) { _dirty_source = src; }
  void set_resident_map(bool on) { _mirror->set_resident(on); }
  const HipMapMirror &mirror() const { return *_mirror; }
  // what init_spe hands WeightedMeanPointProbabilitySPE (slam/scmtch/spe/wmpp/sp_skip_rate, sp_max_usable_range) and
  // whether the map's has_cell() tests a window (PlainGridMap / LazyTiledGridMap): with these the matcher filters the
  // raw scan through slamhip_scan_filter_upload itself whenever no observer wants the filtered LaserScan2D
  void set_filter_params(unsigned skip_rate, double max_range, bool bounded) {
    _skip_rate = skip_rate;
    _max_range = max_range;
    _bounded = bounded;
    _own_filter = true;
  }

  double process_scan(const TransformedLaserScan &raw_scan, const RobotPose &init_pose,
                      const GridMap &map, RobotPoseDelta &pose_delta) override {
    int n_obs = 0;
    do_for_each_observer([&](ObsPtr) { ++n_obs; });
    // (the fast path restates WeightedMeanPointProbabilitySPE's filter_scan / should_skip_point inside the library: it is
    // only taken when the estimator IS that class -- a subclass may override either -- and it leaves no stale _scan)
    if (_own_filter && n_obs == 0 && spe_is_plain_wmpp()) {
      _scan = LaserScan2D{};
      return process_raw_scan(raw_scan, init_pose, map, pose_delta);
    }
    do_for_each_observer([&](ObsPtr obs) { obs->on_matching_start(init_pose, raw_scan, map); });
    _scan = filter_scan(raw_scan.scan, init_pose, map);
    _mirror->sync(map, _dirty_source);
    hip_upload_filtered_scan(_ctx, _scan, _weighting, _trig);
    slamhip_observer o{this, &on_test, &on_update, nullptr};
    slamhip_or_die(slamhip_matcher_set_observer(_m, &o), "set_observer");
    const double p0[3] = {init_pose.x, init_pose.y, init_pose.theta};
    double d[3], prob = 0;
    match_or_unknown(p0, d, &prob);
    pose_delta = RobotPoseDelta{d[0], d[1], d[2]};
    do_for_each_observer([&](ObsPtr obs) { obs->on_matching_end(pose_delta, _scan, prob); });
    return prob;
  }
  long run_time_failures() const { return _n_failures; }

private:
  // A RUN-TIME failure inside a match must not end the SLAM process: a device error or
  // a state error of the launch -- not a configuration error, which stays print + std::exit(-1) like
  // init_scan_matching.h:39-43 -- is retried ONCE on the other device form of the chain (a kernel per super-step
  // instead of the co-resident launch), and if that fails too the match reports what the reference reports when it
  // cannot tell: the "unknown" probability -- a quiet NaN, weighted_mean_point_probability_spe.h:127-131 -- and a
  // zero correction, so the world keeps the odometry pose for this scan.  There is no CPU scorer to fall back on.
  void match_or_unknown(const double p0[3], double d[3], double *prob) {
    int rc = slamhip_matcher_process_scan(_m, _mirror->id(), p0, d, prob);
    if (rc == SLAMHIP_OK) return;
    if (rc == SLAMHIP_ERR_INVALID || rc == SLAMHIP_ERR_UNSUPPORTED || rc == SLAMHIP_ERR_NO_DEVICE)
      slamhip_or_die(rc, "process_scan");
    ++_n_failures;
    std::cerr << "[slamhip] process_scan failed (" << slamhip_last_error() << "): once more on the chain of kernels"
              << std::endl;
    slamhip_matcher_set_device_chain(_m, 1, 0);
    rc = slamhip_matcher_process_scan(_m, _mirror->id(), p0, d, prob);
    slamhip_matcher_set_device_chain(_m, 2, 0);  // (the default form again for the next scan)
    if (rc == SLAMHIP_OK) return;
    ++_n_failures;
    std::cerr << "[slamhip] process_scan failed again (" << slamhip_last_error()
              << "): this scan's match is reported as unknown (NaN, no pose correction)" << std::endl;
    d[0] = d[1] = d[2] = 0.0;
    *prob = std::numeric_limits<double>::quiet_NaN();
  }
  long _n_failures = 0;

  bool spe_is_plain_wmpp() const {
    const auto spe = scan_probability_estimator();
    return spe && typeid(*spe) == typeid(WeightedMeanPointProbabilitySPE);
  }

  // Nobody listens: no filtered LaserScan2D has to exist.  The raw points go to the library as three arrays (kept
  // between scans), which filters, weighs and uploads them (slamhip_scan_filter_upload): no libm call per point on an
  // unbounded map, no sincos per beam per scan.
  double process_raw_scan(const TransformedLaserScan &raw_scan, const RobotPose &init_pose, const GridMap &map,
                          RobotPoseDelta &pose_delta) {
    const auto &pts = raw_scan.scan.points();
    const int n = (int)pts.size();
    _mirror->sync(map, _dirty_source);
    _r.resize(n);
    _a.resize(n);
    _f.resize(n);
    _occ.resize(n);
    for (int i = 0; i < n; ++i) {
      _r[i] = pts[i].range();
      _a[i] = pts[i].angle();
      _f[i] = pts[i].factor();
      _occ[i] = pts[i].is_occupied() ? 1 : 0;
    }
    const double p0[3] = {init_pose.x, init_pose.y, init_pose.theta};
    int kept = 0;
    if (n > 0)
      slamhip_or_die(slamhip_scan_filter_upload(_ctx, _mirror->id(), n, _r.data(), _a.data(), _occ.data(), _f.data(),
                                                _trig.mode, _trig.a_min, _trig.a_max, _trig.a_inc, p0, _skip_rate,
                                                _max_range, _bounded ? 1 : 0, _weighting, &kept, nullptr),
                     "scan_filter_upload");
    if (kept == 0) {  // no usable point: the reference's estimate is NaN for every pose, nothing is ever accepted
      pose_delta = RobotPoseDelta{0, 0, 0};
      return std::numeric_limits<double>::quiet_NaN();
    }
    slamhip_or_die(slamhip_matcher_set_observer(_m, nullptr), "set_observer");
    double d[3], prob = 0;
    match_or_unknown(p0, d, &prob);
    pose_delta = RobotPoseDelta{d[0], d[1], d[2]};
    return prob;
  }
  static void on_test(void *self, const double p[3], double score) {
    auto *t = static_cast<HipGridScanMatcher *>(self);
    t->do_for_each_observer([&](ObsPtr obs) { obs->on_scan_test(RobotPose{p[0], p[1], p[2]}, t->_scan, score); });
  }
  static void on_update(void *self, const double p[3], double score) {
    auto *t = static_cast<HipGridScanMatcher *>(self);
    t->do_for_each_observer([&](ObsPtr obs) { obs->on_pose_update(RobotPose{p[0], p[1], p[2]}, t->_scan, score); });
  }
  slamhip_ctx *_ctx;
  slamhip_matcher *_m;
  std::shared_ptr<HipMapMirror> _mirror;
  int _weighting;
  HipScanTrig _trig;
  const HipMirroredGridMap *_dirty_source = nullptr;
  LaserScan2D _scan;
  bool _own_filter = false, _bounded = false;
  unsigned _skip_rate = 0;
  double _max_range = -1;
  std::vector<double> _r, _a, _f;
  std::vector<int> _occ;
};

// ------------------------------------------------------------------------------------------------
class HipScanProbabilityEstimator : public ScanProbabilityEstimator {
public:
  HipScanProbabilityEstimator(std::shared_ptr<WeightedMeanPointProbabilitySPE> host_spe,
                              slamhip_ctx *ctx, slamhip_spe_cfg cfg,
                              std::shared_ptr<HipMapMirror> mirror, int weighting, HipScanTrig trig = {})
      : ScanProbabilityEstimator{host_spe->occupancy_observation_probability_estimator()},
        _host{host_spe}, _ctx{ctx}, _cfg{cfg}, _mirror{mirror}, _weighting{weighting}, _trig{trig} {}

  LaserScan2D filter_scan(const LaserScan2D &scan, const RobotPose &pose, const GridMap &map) override {
    auto filtered = _host->filter_scan(scan, pose, map);
    _mirror->sync(map);
    hip_upload_filtered_scan(_ctx, filtered, _weighting, _trig);
    return filtered;
  }

  // the scan must be the one returned by the last filter_scan (it is what sits in HBM)
  double estimate_scan_probability(const LaserScan2D &, const RobotPose &pose, const GridMap &,
                                   const SPEParams &params) const override {
    slamhip_spe_cfg cfg = _cfg;
    cfg.area[0] = params.sp_analysis_area.bot();
    cfg.area[1] = params.sp_analysis_area.top();
    cfg.area[2] = params.sp_analysis_area.left();
    cfg.area[3] = params.sp_analysis_area.right();
    const double p[3] = {pose.x, pose.y, pose.theta};
    double score = unknown_probability();
    slamhip_or_die(slamhip_score_poses(_ctx, _mirror->id(), &cfg, 1, p, &score), "score_poses");
    return score;
  }

private:
  std::shared_ptr<WeightedMeanPointProbabilitySPE> _host;
  slamhip_ctx *_ctx;
  slamhip_spe_cfg _cfg;
  std::shared_ptr<HipMapMirror> _mirror;
  int _weighting;
  HipScanTrig _trig;
};

// ------------------------------------------------------------------------------------------------
// read-only GridMap over a device window: occupancy() of the cell kinds the path holds -- the payload double of
// OCC cells; for TBM cells the conversion of the cell class the map was configured with (TbmOccConsistentCell /
// TbmUnknownEvenOccCell::tbm2occ, src/core/maps/tbm_grid_cells.h:76-100; a cell that was never updated still
// reports the prototype's Occupancy{0.5, 1}).  Belief masses: slamhip_map_download_window.
class HipResidentMapView : public GridMap {
public:
  class Cell : public GridCell {
  public:
    explicit Cell(double prob = 0.5) : GridCell{Occupancy{prob, 1.0}} {}
    std::unique_ptr<GridCell> clone() const override { return std::make_unique<Cell>(*this); }
    void set(double prob, double qual = 1.0) { _occupancy = Occupancy{prob, qual}; }
  };
  // tbm_kind: 0 = tbm_consistent, 1 = tbm_unknown_even_occ (TBM windows only)
  HipResidentMapView(slamhip_ctx *ctx, int map_id, const GridMapParams &p, double unknown_prob, int tbm_kind = 0)
      : GridMap{std::make_shared<Cell>(unknown_prob), p}, _ctx{ctx}, _id{map_id}, _unknown{unknown_prob},
        _tbm_kind{tbm_kind} {
    refresh_geometry();
  }
  const GridCell &operator[](const Coord &c) const override {
    const int cx = floor_div(c.x), cy = floor_div(c.y);
    const long long key = ((long long)cy << 32) ^ (unsigned)cx;
    auto it = _chunks.find(key);
    if (it == _chunks.end()) {
      std::vector<Cell> cells((size_t)kChunk * kChunk, Cell{_unknown});
      // the part of the chunk that lies inside the window; the rest reads as the unknown cell
      const int x0 = cx * kChunk + _ox, y0 = cy * kChunk + _oy;  // internal
      const int ix0 = std::max(x0, 0), iy0 = std::max(y0, 0);
      const int ix1 = std::min(x0 + kChunk, _w), iy1 = std::min(y0 + kChunk, _h);
      if (ix0 < ix1 && iy0 < iy1) {
        std::vector<double> tmp((size_t)(ix1 - ix0) * (iy1 - iy0) * _stride);
        slamhip_or_die(slamhip_map_download_window(_ctx, _id, ix0, iy0, ix1 - ix0, iy1 - iy0, tmp.data()),
                       "map_download_window");
        for (int y = iy0; y < iy1; ++y)
          for (int x = ix0; x < ix1; ++x)
            set_cell(cells[(size_t)(y - y0) * kChunk + (x - x0)], &tmp[_stride * ((size_t)(y - iy0) * (ix1 - ix0) + (x - ix0))]);
      }
      it = _chunks.emplace(key, std::move(cells)).first;
    }
    return it->second[(size_t)(c.y - cy * kChunk) * kChunk + (c.x - cx * kChunk)];
  }
  void update(const Coord &, const AreaOccupancyObservation &) override {}  // a view: the world writes through K6
  void reset(const Coord &, const GridCell &) override {}
  DiscretePoint2D origin() const override { return DiscretePoint2D{_ox, _oy}; }
  // an unbounded map has every cell (UnboundedPlainGridMap::has_cell, plain_grid_map.h:77) -- filter_scan drops
  // the points whose cell the map does not have (weighted_mean_point_probability_spe.h:135-140)
  bool has_cell(const Coord &) const override { return true; }
  // after an update: cached chunks are stale, the window may have grown
  void invalidate() {
    _chunks.clear();
    refresh_geometry();
  }

private:
  void set_cell(Cell &c, const double *p) const {
    if (_stride != 4) {
      c.set(p[0]);
    } else if (p[0] == 1.0 && p[1] == 0.0 && p[2] == 0.0) {
      c.set(0.5);  // total ignorance: the cell was never updated
    } else if (_tbm_kind == 1) {
      c.set(p[2] + 0.5 * p[0]);
    } else {
      const double qual = p[2] + p[1];
      c.set(p[2] / qual, qual);
    }
  }
  void refresh_geometry() {
    int model = 0;
    slamhip_or_die(slamhip_map_info(_ctx, _id, &model, &_w, &_h, &_ox, &_oy, nullptr, nullptr), "map_info");
    _stride = model == SLAMHIP_CELL_TBM ? 4 : (model == SLAMHIP_CELL_GMAPPING ? 3 : 1);
    set_width(_w);
    set_height(_h);
  }
  static constexpr int kChunk = 64;
  static int floor_div(int v) { return v >= 0 ? v / kChunk : -((-v + kChunk - 1) / kChunk); }
  slamhip_ctx *_ctx;
  int _id;
  double _unknown;
  int _tbm_kind;
  int _w = 0, _h = 0, _ox = 0, _oy = 0, _stride = 1;
  mutable std::unordered_map<long long, std::vector<Cell>> _chunks;
};

#endif  // SLAMHIP_REFERENCE_ADAPTER_H
