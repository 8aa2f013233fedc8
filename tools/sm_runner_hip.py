#!/usr/bin/env python3
"""sm_runner over the HIP engine: the reference's offline scan-matching tool
(src/utils/sm_runner.cpp:64-96) with the matcher swapped for slamhip.

  sm_runner_hip.py <config.properties> <file.pose2D> <file.map> <file.scan2D>

Reads the same four fixture files (formats: slam-constructor_amd/fixtures.py), builds the matcher
the properties describe (init_scan_matching.h:24-229), and prints what the reference prints, ending
with "Pose delta: PoseDelta{ x: .., y: .., th: ..} with probability p".  Needs a GPU: there is no
CPU fallback.
"""
import importlib.util
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pkg():
    spec = importlib.util.spec_from_file_location("graft_entry", os.path.join(ROOT, "__graft_entry__.py"))
    ge = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ge)
    return ge.load_package()


def _bool(v):  # MapPropertiesProvider::get_bool (properties_providers.h:66-70)
    return not (v == "false" or v == "0")


class Props:
    def __init__(self, d):
        self.d = d

    def s(self, k, dflt):
        return self.d.get(k, dflt)

    def f(self, k, dflt):
        return float(self.d[k]) if k in self.d else dflt

    def i(self, k, dflt):
        return int(self.d[k]) if k in self.d else dflt

    def b(self, k, dflt):
        return _bool(self.d[k]) if k in self.d else dflt


SM = "slam/scmtch/"


def describe(props):
    """properties -> (cell kind, bounded, spe kwargs, weighting, skip, max_range, matcher kind, params, log)"""
    log = []
    area = props.s("slam/mapping/grid/area/type", "<undefined>")
    if area in ("tbm_consistent", "tbm_unknown_even_occ"):
        cell = "tbm"
    elif area in ("affine_quality_merge", "mean_probability"):
        cell = "base"
    else:
        raise SystemExit("Unknown occupied area type: " + area)
    gm = props.s("slam/mapping/grid/type", "<undefined>")
    if gm not in ("plain", "unbounded_plain", "lazy_tiled", "unbounded_lazy_tiled"):
        raise SystemExit("Unknown grid map type (slam/mapping/grid/type): " + gm)
    if gm != "unbounded_plain":
        # only UnboundedPlainGridMap implements load_state (plain_grid_map.h:100-129); the other
        # types keep GridMap's no-op (grid_map.h:64) and the reference tool then matches on an
        # empty map -- refuse rather than mimic that
        raise SystemExit("grid map type %s has no load_state in the reference: use unbounded_plain" % gm)
    bounded = False

    oie = props.s(SM + "oie/type", "discrepancy")
    oope = props.s(SM + "oope/type", "obstacle")
    log += ["Used OIE: " + oie, "Used OOPE: " + oope]
    if oie not in ("discrepancy", "occupancy"):
        raise SystemExit("Unknown observation impact estimator type (%soie/) %s" % (SM, oie))
    if oope not in ("obstacle", "max", "mean", "overlap"):
        raise SystemExit("Unknown occupancy observation probability estimator type (%soope/type) %s" % (SM, oope))
    if props.s(SM + "spe/type", "<undefined>") != "wmpp":
        raise SystemExit("Unknown Scan Probability Estimator type (%sspe/type): %s"
                         % (SM, props.s(SM + "spe/type", "<undefined>")))
    skip = props.i(SM + "spe/wmpp/sp_skip_rate", 0)
    max_range = props.f(SM + "spe/wmpp/sp_max_usable_range", -1.0)
    swp = props.s(SM + "spe/wmpp/weighting/type", "<undefined>")
    log.append("Used SWP: " + swp)
    if swp not in ("even", "viny", "ahr"):
        raise SystemExit("Unknown Scan Point Weighting type (%sspe/wmpp/weighting/type) %s" % (SM, swp))

    kind = props.s(SM + "type", "<undefined>")
    log.append("Used Scan Matcher: " + kind)
    if kind == "MC":
        ns = SM + "MC/"
        if ns + "seed" not in props.d:
            raise SystemExit("MC without %sseed draws a random_device seed: not reproducible, refused" % ns)
        seed = props.i(ns + "seed", 0)
        log.append("[INFO] MC Scan Matcher seed: %d" % seed)
        params = (seed, props.f(ns + "dispersion/translation", 0.2), props.f(ns + "dispersion/rotation", 0.1),
                  props.i(ns + "dispersion/failed_attempts_limit", 20), props.i(ns + "attempts_limit", 100))
    elif kind == "HC":
        ns = SM + "HC/distortion/"
        if props.b(SM + "HC/use_frame_alignement", False):
            # HillClimbingScanMatcher's 5th constructor argument lands in a bool that process_scan
            # never reads (hill_climbing_scan_matcher.h:137-143): accepted, no effect
            pass
        params = (props.i(ns + "failed_attempts_limit", 6), props.f(ns + "translation", 0.1),
                  props.f(ns + "rotation", 0.1))
    elif kind == "BF":
        ns = SM + "BF/"
        params = []
        for dim, lim, step in (("x", 0.5, 0.1), ("y", 0.5, 0.1), ("t", math.radians(5), math.radians(1))):
            params += [props.f(ns + dim + "/from", -lim), props.f(ns + dim + "/to", lim),
                       props.f(ns + dim + "/step", step)]
    else:
        raise SystemExit("scan matcher type %r is outside the HIP path (MC / HC / BF)" % kind)
    if props.b(SM + "use_amb_drift_detector", False):
        raise SystemExit("use_amb_drift_detector wraps the matcher in host-only code: out of scope")
    return dict(cell=cell, bounded=bounded, oie=oie, oope=oope, swp=swp, skip=skip, max_range=max_range,
                kind=kind, params=params, log=log)


def run(cfg_path, pose_path, map_path, scan_path, out=sys.stdout, strict=False):
    sh = _pkg()
    from importlib import import_module
    fx = import_module(sh.__name__ + ".fixtures")
    props = Props(fx.read_properties(cfg_path))
    d = describe(props)
    pose = fx.read_pose2d(pose_path)
    rng, ang, occ = fx.read_scan2d(scan_path)
    m = fx.read_map(map_path, d["cell"])
    m.bounded = d["bounded"]

    # LaserScan2D{} carries a RawTrigonometryProvider (sensor_data.h:152)
    geom = dict(width=m.width, height=m.height, origin=m.origin, scale=m.scale, bounded=m.bounded)
    kept = sh.filter_scan(rng, ang, occ, pose, geom, skip_rate=d["skip"], max_range=d["max_range"])
    f_rng, f_ang = rng[kept], ang[kept]
    weight = sh.scan_weights(d["swp"], f_rng, f_ang)
    cos_a, sin_a = sh.beam_trig(f_ang)

    ctx = sh.Context(0)
    ctx.upload_map(0, m)
    ctx.scan_upload(f_rng, cos_a, sin_a, weight)
    oope = {"obstacle": sh.OOPE_OBSTACLE, "max": sh.OOPE_MAX, "mean": sh.OOPE_MEAN, "overlap": sh.OOPE_OVERLAP}
    oie = {"discrepancy": sh.OIE_DISCREPANCY, "occupancy": sh.OIE_OCCUPANCY}
    kw = dict(sum_order=sh.SUM_SEQUENTIAL, pose_trig=sh.POSE_TRIG_HOST) if strict else {}
    cfg = sh.spe_cfg(oope=oope[d["oope"]], oie=oie[d["oie"]], **kw)
    matcher = sh.Matcher(ctx, d["kind"], cfg, d["params"])
    res = matcher.process_scan(0, pose)
    for line in d["log"]:
        print(line, file=out)
    print("Pose delta: PoseDelta{ x: %g, y: %g, th: %g} with probability %g"
          % (res["delta"][0], res["delta"][1], res["delta"][2], res["prob"]), file=out)
    return res


def main(argv):
    if len(argv) != 5:
        print("Usage: sm_runner_hip.py <config.properties> <file.pose2D> <file.map> <file.scan2D>")
        return -1
    run(*argv[1:5], strict=os.environ.get("SLAMHIP_STRICT", "0") == "1")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
