// slamhip_init_slam.h -- the factory a maintainer adds next to init_1h_slam (src/utils/init_slam.h:12-25):
// the same single-hypothesis world (tinySLAM / vinySLAM presets), built from the same properties, whose
// scan matcher runs on the GPU.  The world class, the map class and the scan adder stay the
// reference's own; the map is wrapped in HipMirroredGridMap so that the cells
// GridMapScanAdder::append_scan updates after every match
// (src/core/states/single_state_hypothesis_laser_scan_grid_world.h:52-65) reach the HBM window as a
// dirty-cell log before the next match.  Compiled only with the reference headers on the include path;
// contains no reference code.  oracle/ref_world_harness.cpp runs it next to init_1h_slam.
#ifndef SLAMHIP_INIT_SLAM_H
#define SLAMHIP_INIT_SLAM_H

#include <memory>
#include <tuple>

#include "utils/init_slam.h"
#include "slamhip_init_scan_matching.h"

// wrap_map = false keeps the reference's bare map object: the mirror then compares the whole map with
// its host shadow before every match (correct, one virtual call per cell and scan)
inline auto init_hip_1h_slam(const PropertiesProvider &props, slamhip_ctx *ctx = nullptr, int map_id = 0,
                             bool wrap_map = true) {
  auto slam_props = SingleStateHypothesisLSGWProperties{};
  double loc, raw;
  std::tie(loc, raw) = init_pose_quality_estimators(props);
  slam_props.localized_scan_quality = loc;
  slam_props.raw_scan_quality = raw;
  std::shared_ptr<GridMap> map = init_grid_map(props);
  if (wrap_map) map = std::make_shared<HipMirroredGridMap>(map);
  slam_props.grid_map = map;
  slam_props.gsm = init_hip_scan_matcher(props, ctx, map_id);
  slam_props.gmsa = init_scan_adder(props);
  return std::make_shared<SingleStateHypothesisLaserScanGridWorld>(slam_props);
}

#endif  // SLAMHIP_INIT_SLAM_H
