// slamhip_gmapping_adapter.h -- reference-side binding of the GMapping particle-filter path.
//
// Injection point (3) of SURVEY 8b: the LaserScanGridWorld object handed to the LaserScanObserver
// (src/slams/gmapping/gmapping.cpp:22,45-48; src/ros/lslam2D_bag_runner.cpp:164-166).  Like
// slamhip_reference_adapter.h this header is compiled only with the reference headers on the include
// path, contains no reference code and is not part of libslamhip.so.
//
//   HipDeviceGridMap           a read-only GridMap view of a map that lives in HBM: operator[] fetches
//                              64 x 64-cell chunks on demand (slamhip_map_download_window or
//                              slamhip_gmapping_particle_map_download) and keeps them until the next
//                              filter step, so handing `map()` to observers costs nothing unless
//                              somebody reads cells (the map publisher does, every few seconds)
//   HipGmappingParticleFilter  LaserScanGridWorld with GmappingParticleFilter's behaviour
//                              (src/slams/gmapping/gmapping_particle_filter.h:29-118): handle_sensor_data
//                              = slamhip_gmapping_step, pose() / map() = the heaviest particle's
//                              (particle_filter.h:114-121).  Map modes: the reference's shared map
//                              with the update inside the step (default), or per-particle
//                              copy-on-write maps.
#ifndef SLAMHIP_GMAPPING_ADAPTER_H
#define SLAMHIP_GMAPPING_ADAPTER_H

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <iostream>
#include <memory>
#include <random>
#include <unordered_map>
#include <vector>

#include "core/maps/grid_map.h"
#include "core/states/laser_scan_grid_world.h"
#include "slamhip.h"

#ifndef SLAMHIP_REFERENCE_ADAPTER_H
inline void slamhip_or_die(int rc, const char *what) {
  if (rc == SLAMHIP_OK) return;
  std::cerr << "[slamhip] " << what << ": " << slamhip_last_error() << std::endl;
  std::exit(-1);
}
#endif

// occupancy-only cell of the host view (what map consumers read: GridMap::occupancy,
// GridCell::occupancy; src/core/maps/grid_cell.h:15-35)
class HipViewCell : public GridCell {
public:
  explicit HipViewCell(double prob = 0.5) : GridCell{Occupancy{prob, 1.0}} {}
  std::unique_ptr<GridCell> clone() const override { return std::make_unique<HipViewCell>(*this); }
  void set(double prob) { _occupancy = Occupancy{prob, 1.0}; }
};

class HipDeviceGridMap : public GridMap {
public:
  // fetch(x0, y0, w, h, payload): external window -> stride-3 payload (prob_occ first)
  using Fetch = std::function<void(int, int, int, int, double *)>;
  HipDeviceGridMap(const GridMapParams &p, DiscretePoint2D origin, double unknown_prob, Fetch fetch)
      : GridMap{std::make_shared<HipViewCell>(unknown_prob), p}, _origin{origin}, _fetch{std::move(fetch)},
        _unknown{unknown_prob} {}

  const GridCell &operator[](const Coord &c) const override {
    const int cx = floor_div(c.x), cy = floor_div(c.y);
    const long long key = ((long long)cy << 32) ^ (unsigned)cx;
    auto it = _chunks.find(key);
    if (it == _chunks.end()) {
      std::vector<double> buf((size_t)kChunk * kChunk * 3);
      _fetch(cx * kChunk, cy * kChunk, kChunk, kChunk, buf.data());
      std::vector<HipViewCell> cells((size_t)kChunk * kChunk, HipViewCell{_unknown});
      for (size_t i = 0; i < cells.size(); ++i) cells[i].set(buf[3 * i]);
      it = _chunks.emplace(key, std::move(cells)).first;
    }
    return it->second[(size_t)(c.y - cy * kChunk) * kChunk + (c.x - cx * kChunk)];
  }
  void update(const Coord &, const AreaOccupancyObservation &) override {}  // a view: the filter writes
  void reset(const Coord &, const GridCell &) override {}
  DiscretePoint2D origin() const override { return _origin; }
  bool has_cell(const Coord &) const override { return true; }
  void invalidate() const { _chunks.clear(); }
  // the device window grew (slamhip_map_set_auto_grow): RegularSquaresGrid::width / height / origin follow it
  void set_geometry(DiscretePoint2D origin, int w, int h) {
    _origin = origin;
    set_width(w);
    set_height(h);
  }

private:
  static constexpr int kChunk = 64;
  static int floor_div(int v) { return v >= 0 ? v / kChunk : -((-v + kChunk - 1) / kChunk); }
  DiscretePoint2D _origin;
  Fetch _fetch;
  double _unknown;
  mutable std::unordered_map<long long, std::vector<HipViewCell>> _chunks;
};

class HipGmappingParticleFilter : public LaserScanGridWorld {
public:
  static constexpr double kUnknownProb = -1.0;
  struct Config {
    slamhip_gmapping_params filter{};       // init_gmapping_params + HC(6, 0.1, 0.1) + SPE + OOPE
    slamhip_scan_adder_cfg adder{};         // init_scan_adder (init_occupancy_mapping.h:82-92)
    GridMapParams map{1000, 1000, 0.1};     // init_grid_map_params
    bool particle_maps = false;             // false: the reference's one shared map (Q20)
    int extent_tiles = 64, pool_tiles = 4096;  // per-particle maps only
    int map_id = 0;
    // Sharded filter: when the context has joined an RCCL group (slamhip_shard_init) this object holds
    // the particles [first, first + count) of n -- contiguous blocks in rank order, the first n % world
    // ranks one more -- and every step is slamhip_gmapping_step_sharded: lock-step matching of the
    // local block, ONE all-gather of the raw weights, identical resampling on every rank.  The map is
    // replicated and not updated inside the step (the reference's particles write ONE shared map one
    // after the other, Q20: that step cannot be split).
  };
  // seed_source: what std::random_device is to the reference (GmappingWorld ctor, gmapping_world.h:51;
  // UniformResamling::resample, particle_filter.h:51-52): n draws now, one per step
  HipGmappingParticleFilter(slamhip_ctx *ctx, const Config &cfg, unsigned n,
                            std::function<unsigned()> seed_source = std::random_device{})
      : _ctx{ctx}, _cfg{cfg}, _n{n}, _seed{std::move(seed_source)} {
    // GmappingBaseCell prototype: Occupancy{-1, 1}, obstacle (0, 0) (gmapping_grid_cell.h:14)
    const double unknown[4] = {kUnknownProb, 0.0, 0.0, 0.0};
    const int w = cfg.map.width_cells, h = cfg.map.height_cells;
    slamhip_or_die(slamhip_map_bind(ctx, cfg.map_id, SLAMHIP_CELL_GMAPPING, w, h, w / 2, h / 2,
                                    cfg.map.meters_per_cell, unknown), "map_bind");
    // the reference's map is unbounded (UnboundedLazyTiledGridMap, init_gmapping.h:27-33): a scan that reaches
    // beyond the window makes the shared dense window grow instead of failing the step
    slamhip_or_die(slamhip_map_set_auto_grow(ctx, cfg.map_id, 1), "map_set_auto_grow");
    std::vector<uint32_t> seeds(n);
    for (auto &s : seeds) s = _seed();
    slamhip_or_die(slamhip_shard_info(ctx, &_rank, &_world), "shard_info");
    if (_world > 1) {
      if (cfg.particle_maps) {
        std::cerr << "[slamhip] a sharded HipGmappingParticleFilter keeps one replicated map" << std::endl;
        std::exit(-1);
      }
      _counts.resize(_world);
      for (int r = 0; r < _world; ++r) _counts[r] = (int)n / _world + (r < (int)n % _world ? 1 : 0);
      _first = 0;
      for (int r = 0; r < _rank; ++r) _first += _counts[r];
      slamhip_or_die(slamhip_gmapping_create(ctx, &cfg.filter, (int)n, _first, _counts[_rank], seeds.data() + _first,
                                             &_pf), "gmapping_create");
    } else {
      slamhip_or_die(slamhip_gmapping_create(ctx, &cfg.filter, (int)n, 0, (int)n, seeds.data(), &_pf), "gmapping_create");
    }
    if (_world > 1) {
      // likelihood step only, see Config
    } else if (cfg.particle_maps)
      slamhip_or_die(slamhip_gmapping_enable_particle_maps(_pf, cfg.map_id, &cfg.adder, cfg.extent_tiles,
                                                           cfg.pool_tiles), "enable_particle_maps");
    else
      slamhip_or_die(slamhip_gmapping_set_map_update(_pf, &cfg.adder), "set_map_update");
    _poses.assign(3 * n, 0.0);
    _weights.assign(n, 1.0 / n);
    _heaviest = n - 1;
    _view = std::make_shared<HipDeviceGridMap>(
        cfg.map, DiscretePoint2D{w / 2, h / 2}, kUnknownProb, [this](int x0, int y0, int ww, int hh, double *out) {
          if (_cfg.particle_maps) {
            slamhip_or_die(slamhip_gmapping_particle_map_download(_pf, (int)_heaviest, x0, y0, ww, hh, out, nullptr),
                           "particle_map_download");
            return;
          }
          // dense window (it grows with the scans, see below): clip to what is bound, the rest reads as unknown
          int mw = 0, mh = 0, ox = 0, oy = 0;
          slamhip_or_die(slamhip_map_info(_ctx, _cfg.map_id, nullptr, &mw, &mh, &ox, &oy, nullptr, nullptr), "map_info");
          for (size_t i = 0; i < (size_t)ww * hh; ++i) out[3 * i] = kUnknownProb, out[3 * i + 1] = out[3 * i + 2] = 0.0;
          const int ix0 = std::max(x0 + ox, 0), iy0 = std::max(y0 + oy, 0);
          const int ix1 = std::min(x0 + ox + ww, mw), iy1 = std::min(y0 + oy + hh, mh);
          if (ix0 >= ix1 || iy0 >= iy1) return;
          std::vector<double> tmp((size_t)(ix1 - ix0) * (iy1 - iy0) * 3);
          slamhip_or_die(slamhip_map_download_window(_ctx, _cfg.map_id, ix0, iy0, ix1 - ix0, iy1 - iy0, tmp.data()),
                         "map_download_window");
          for (int y = iy0; y < iy1; ++y)
            for (int x = ix0; x < ix1; ++x)
              for (int k = 0; k < 3; ++k)
                out[3 * ((size_t)(y - oy - y0) * ww + (x - ox - x0)) + k] =
                    tmp[3 * ((size_t)(y - iy0) * (ix1 - ix0) + (x - ix0)) + k];
        });
  }
  ~HipGmappingParticleFilter() override { slamhip_gmapping_destroy(_pf); }

  void handle_sensor_data(TransformedLaserScan &scan) override {
    update_robot_pose(scan.pose_delta);
    handle_observation(scan);
    notify_with_pose(pose());
    notify_with_map(map());
  }
  // the odometry update is part of the step (GmappingWorld::update_robot_pose rotates the delta by
  // every particle's own heading correction, gmapping_world.h:57-71)
  void update_robot_pose(const RobotPoseDelta &) override {}

  const RobotPose &pose() const override { return _pose; }
  const MapType &map() const override { return *_view; }

  // test / diagnostics access
  const std::vector<double> &particle_poses() const { return _poses; }
  const std::vector<double> &particle_weights() const { return _weights; }
  bool resampled_last_step() const { return _resampled; }
  slamhip_gmapping *handle() { return _pf; }

protected:
  void handle_observation(TransformedLaserScan &obs) override {
    const auto &pts = obs.scan.points();
    const size_t m = pts.size();
    _range.resize(m);
    _angle.resize(m);
    _occ.resize(m);
    for (size_t i = 0; i < m; ++i) {
      _range[i] = pts[i].range();
      _angle[i] = pts[i].angle();
      _occ[i] = pts[i].is_occupied() ? 1 : 0;
    }
    const double d[3] = {obs.pose_delta.x, obs.pose_delta.y, obs.pose_delta.theta};
    int res = 0;
    if (_world > 1) {
      // the resampling seed has to be the same everywhere: rank 0's draw
      std::vector<unsigned> draws(_world);
      const std::vector<int> ones(_world, 1);
      const unsigned mine = _seed();
      slamhip_or_die(slamhip_shard_allgather(_ctx, &mine, ones.data(), (int)sizeof(unsigned), draws.data()), "seed");
      slamhip_or_die(slamhip_gmapping_step_sharded(_pf, _cfg.map_id, (int)m, _range.data(), _angle.data(), _occ.data(),
                                                   d, draws[0], &res, nullptr), "gmapping_step_sharded");
      std::vector<double> lp(3 * (size_t)_counts[_rank]), lw(_counts[_rank]);
      slamhip_or_die(slamhip_gmapping_get(_pf, lp.data(), lw.data(), nullptr), "gmapping_get");
      slamhip_or_die(slamhip_shard_allgather(_ctx, lp.data(), _counts.data(), 3 * (int)sizeof(double), _poses.data()),
                     "poses");
      slamhip_or_die(slamhip_shard_allgather(_ctx, lw.data(), _counts.data(), (int)sizeof(double), _weights.data()),
                     "weights");
    } else {
      slamhip_or_die(slamhip_gmapping_step(_pf, _cfg.map_id, (int)m, _range.data(), _angle.data(), _occ.data(), d,
                                           _seed(), &res, nullptr), "gmapping_step");
      slamhip_or_die(slamhip_gmapping_get(_pf, _poses.data(), _weights.data(), nullptr), "gmapping_get");
    }
    _resampled = res != 0;
    // heaviest_particle: the LAST of equal maxima (particle_filter.h:114-121)
    _heaviest = 0;
    for (unsigned i = 0; i < _n; ++i)
      if (!(_weights[i] < _weights[_heaviest])) _heaviest = i;
    _pose = RobotPose{_poses[3 * _heaviest], _poses[3 * _heaviest + 1], _poses[3 * _heaviest + 2]};
    _view->invalidate();
    if (!_cfg.particle_maps) {
      int mw = 0, mh = 0, ox = 0, oy = 0;
      slamhip_or_die(slamhip_map_info(_ctx, _cfg.map_id, nullptr, &mw, &mh, &ox, &oy, nullptr, nullptr), "map_info");
      _view->set_geometry(DiscretePoint2D{ox, oy}, mw, mh);
    }
  }

private:
  slamhip_ctx *_ctx;
  Config _cfg;
  unsigned _n;
  std::function<unsigned()> _seed;
  slamhip_gmapping *_pf = nullptr;
  std::vector<double> _poses, _weights, _range, _angle;
  std::vector<int> _occ;
  unsigned _heaviest = 0;
  bool _resampled = false;
  int _rank = 0, _world = 1, _first = 0;
  std::vector<int> _counts;
  RobotPose _pose{0, 0, 0};
  std::shared_ptr<HipDeviceGridMap> _view;
};

#endif  // SLAMHIP_GMAPPING_ADAPTER_H
