// slamhip_init_scan_matching.h -- the factory a maintainer adds next to init_scan_matcher
// (src/utils/init_scan_matching.h:193-218): same properties, same console lines, but the matcher it
// returns runs on the GPU through the C-ABI.  Compiled only with the reference headers on the include
// path; contains no reference code.  oracle/sm_runner_hip.cpp is the reference's offline tool
// (src/utils/sm_runner.cpp) with this factory in place of the reference's.
#ifndef SLAMHIP_INIT_SCAN_MATCHING_H
#define SLAMHIP_INIT_SCAN_MATCHING_H

#include <cstdlib>
#include <iostream>
#include <memory>
#include <string>

#include "utils/init_scan_matching.h"
#include "slamhip_reference_adapter.h"

inline std::shared_ptr<GridScanMatcher> init_hip_scan_matcher(const PropertiesProvider &props,
                                                              slamhip_ctx *ctx = nullptr, int map_id = 0) {
  if (!ctx) slamhip_or_die(slamhip_ctx_create(props.get_int("slam/scmtch/hip/device", 0), &ctx), "ctx_create");
  // the reference's own SPE: still does filter_scan on the host and prints "Used OIE/OOPE/SWP"
  auto spe = init_spe(props);
  slamhip_spe_cfg cfg{};
  const auto oope = props.get_str(Slam_SM_NS + "oope/type", "obstacle");
  cfg.oope = oope == "max" ? SLAMHIP_OOPE_MAX
             : oope == "mean" ? SLAMHIP_OOPE_MEAN
             : oope == "overlap" ? SLAMHIP_OOPE_OVERLAP : SLAMHIP_OOPE_OBSTACLE;
  cfg.oie = props.get_str(Slam_SM_NS + "oie/type", "discrepancy") == "occupancy" ? SLAMHIP_OIE_OCCUPANCY
                                                                                 : SLAMHIP_OIE_DISCREPANCY;
  cfg.sum_order = props.get_bool(Slam_SM_NS + "hip/strict", false) ? SLAMHIP_SUM_SEQUENTIAL : SLAMHIP_SUM_TREE256;
  // strict: the reference's bits.  Scans carry the RawTrigonometryProvider unless the node was started with
  // use_trig_cache (src/ros/init_utils.h:56-58; laser_scan_observer.h:77-86) -- "hip/trig_cache" says so here --:
  // cos / sin(theta + a) per beam by the restated libm (RAW_EXACT), or the cached provider's angle addition (HOST)
  const bool strict = props.get_bool(Slam_SM_NS + "hip/strict", false);
  const bool trig_cache = props.get_bool(Slam_SM_NS + "hip/trig_cache", false);
  cfg.pose_trig = !strict ? SLAMHIP_POSE_TRIG_DEVICE : (trig_cache ? SLAMHIP_POSE_TRIG_HOST : SLAMHIP_POSE_TRIG_RAW_EXACT);
  const auto type = scan_matcher_type(props);
  std::cout << "Used Scan Matcher: " << type << std::endl;
  slamhip_matcher *m = nullptr;
  if (type == "MC") {
    const std::string ns = Slam_SM_NS + "MC/";
    const auto seed = props.get_int(ns + "seed", std::random_device{}());
    std::cout << "[INFO] MC Scan Matcher seed: " << seed << std::endl;
    slamhip_or_die(slamhip_matcher_create_mc(ctx, &cfg, seed, props.get_dbl(ns + "dispersion/translation", 0.2),
                                             props.get_dbl(ns + "dispersion/rotation", 0.1),
                                             props.get_uint(ns + "dispersion/failed_attempts_limit", 20),
                                             props.get_uint(ns + "attempts_limit", 100), &m), "create_mc");
  } else if (type == "HC") {
    const std::string ns = Slam_SM_NS + "HC/distortion/";
    slamhip_or_die(slamhip_matcher_create_hc(ctx, &cfg, props.get_uint(ns + "failed_attempts_limit", 6),
                                             props.get_dbl(ns + "translation", 0.1),
                                             props.get_dbl(ns + "rotation", 0.1), &m), "create_hc");
  } else if (type == "BF") {
    const std::string ns = Slam_SM_NS + "BF/";
    const double r[9] = {props.get_dbl(ns + "x/from", -0.5), props.get_dbl(ns + "x/to", 0.5),
                         props.get_dbl(ns + "x/step", 0.1), props.get_dbl(ns + "y/from", -0.5),
                         props.get_dbl(ns + "y/to", 0.5), props.get_dbl(ns + "y/step", 0.1),
                         props.get_dbl(ns + "t/from", -deg2rad(5)), props.get_dbl(ns + "t/to", deg2rad(5)),
                         props.get_dbl(ns + "t/step", deg2rad(1))};
    slamhip_or_die(slamhip_matcher_create_bf(ctx, &cfg, r, &m), "create_bf");
  } else {
    std::cerr << "scan matcher type " << type << " is outside the HIP path (MC / HC / BF)" << std::endl;
    std::exit(-1);
  }
  const auto area = props.get_str("slam/mapping/grid/area/type", "<undefined>");
  // TBM cells score through their belief masses; every other model (and the occupancy OIE) through
  // occupancy().prob_occ
  const int model = (area.rfind("tbm", 0) == 0 && cfg.oie == SLAMHIP_OIE_DISCREPANCY) ? SLAMHIP_CELL_TBM
                                                                                        : SLAMHIP_CELL_OCC;
  const auto w = props.get_str(Slam_SM_NS + "spe/wmpp/weighting/type", "even");
  const int weighting = w == "viny" ? 1 : (w == "ahr" ? 2 : 0);
  const auto gm = props.get_str("slam/mapping/grid/type", "<undefined>");
  const bool bounded = gm == "plain" || gm == "lazy_tiled";
  auto mirror = std::make_shared<HipMapMirror>(ctx, map_id, model, bounded);
  auto gsm = std::make_shared<HipGridScanMatcher>(spe, ctx, m, mirror, weighting);
  if (props.get_str(Slam_SM_NS + "spe/type", "<undefined>") == "wmpp")  // (init_spe's own parameters, :99-100)
    gsm->set_filter_params(props.get_uint(Slam_SM_NS + "spe/wmpp/sp_skip_rate", 0),
                           props.get_dbl(Slam_SM_NS + "spe/wmpp/sp_max_usable_range", -1), bounded);
  return gsm;
}

#endif  // SLAMHIP_INIT_SCAN_MATCHING_H
