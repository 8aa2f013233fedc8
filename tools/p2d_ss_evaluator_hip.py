#!/usr/bin/env python3
"""pose2D search-space evaluator over the HIP engine.

What src/utils/pose2D_search_space_evaluator.cpp:154-184 means to do for a scene: sweep
BruteForceScanMatcher(-1..1 m step `resolution` in x and y, no rotation) around the true pose with
an observer that writes every tested pose's score into an UnboundedPlainGridMap of cell size
`resolution` (ScanMatcherSearchSpaceBuilder, :32-61), dump that map as `sss_map_0.pgm`, the input map
as `input_map_0.pgm`, and print "BF: <seconds>" -- the only timing print in the reference.  (The
reference binary itself crashes in dump_scan before evaluating anything, see
tests/golden/make_golden_search_space.py; the scenes are therefore taken from a fixture built call
by call through the compiled reference.)

    p2d_ss_evaluator_hip.py [--fixture tests/golden/search_space.npz] [--scene closed|open|several|all]
                            [--out DIR]

Needs a GPU: there is no CPU fallback.
"""
import argparse
import importlib.util
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TITLES = {"closed": "[1] Closed Corridor", "open": "[2] Open Corridor", "several": "[3] Several Corridors"}


def _pkg():
    spec = importlib.util.spec_from_file_location("graft_entry", os.path.join(ROOT, "__graft_entry__.py"))
    ge = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ge)
    return ge.load_package()


class Window:
    def __init__(self, payload, origin, scale):
        self.cell_model, self.payload = 0, np.ascontiguousarray(payload, dtype=np.float64)
        self.height, self.width = self.payload.shape[:2]
        self.origin, self.scale = (int(origin[0]), int(origin[1])), float(scale)
        self.unknown = np.array([0.5, 0.0, 0.0, 0.0])


def evaluate(sh, fx, ctx, g, scene, strict=True):
    """-> dict(scores, poses, prob, delta, n_calls, seconds, sss_prob, sss_geometry)"""
    m = Window(g[scene + "_map_payload"], g[scene + "_map_origin"], 0.1)
    rng, ang, occ = g[scene + "_scan"]
    pose, res = g["pose"], float(g["resolution"])
    geom = dict(width=m.width, height=m.height, origin=m.origin, scale=m.scale, bounded=False)
    kept = sh.filter_scan(rng, ang, occ.astype(np.int32), pose, geom)
    f_rng, f_ang = rng[kept], ang[kept]
    cos_a, sin_a = sh.beam_trig(f_ang)
    ctx.upload_map(0, m)
    ctx.scan_upload(f_rng, cos_a, sin_a, sh.scan_weights("even", f_rng, f_ang))
    kw = dict(sum_order=sh.SUM_SEQUENTIAL, pose_trig=sh.POSE_TRIG_HOST) if strict else {}
    matcher = sh.Matcher(ctx, "BF", sh.spe_cfg(**kw), g["params"])
    t0 = time.perf_counter()
    plain = matcher.process_scan(0, pose)  # what the reference times: the match itself
    seconds = time.perf_counter() - t0
    t = sh.Matcher(ctx, "BF", sh.spe_cfg(**kw), g["params"]).process_scan(0, pose, trace=True)
    assert t["prob"] == plain["prob"] and np.array_equal(t["delta"], plain["delta"])
    # ScanMatcherSearchSpaceBuilder::on_scan_test: last write wins at world_to_cell(pose)
    win = fx.UnboundedWindow(100, 100)
    cx = np.floor(t["poses"][:, 0] / res).astype(np.int64)
    cy = np.floor(t["poses"][:, 1] / res).astype(np.int64)
    cells = {}
    for k in range(t["n_calls"]):
        win.ensure_inside(int(cx[k]), int(cy[k]))
        cells[(int(cx[k]), int(cy[k]))] = t["scores"][k]
    prob = np.full((win.height, win.width), 0.5)
    for (x, y), s in cells.items():
        prob[y + win.origin[1], x + win.origin[0]] = s
    t.update(seconds=seconds, sss_prob=prob, sss_geometry=(win.width, win.height, win.origin[0], win.origin[1]),
             input_prob=m.payload[..., 0], beams=len(kept))
    return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fixture", default=os.path.join(ROOT, "tests", "golden", "search_space.npz"))
    ap.add_argument("--scene", default="all")
    ap.add_argument("--out", default=".")
    ap.add_argument("--default-mode", action="store_true", help="tree sum + device sincos instead of strict")
    args = ap.parse_args()
    sh = _pkg()
    from importlib import import_module
    fx = import_module(sh.__name__ + ".fixtures")
    g = dict(np.load(args.fixture))
    ctx = sh.Context(0)
    os.makedirs(args.out, exist_ok=True)
    for scene in (["closed", "open", "several"] if args.scene == "all" else [args.scene]):
        print(TITLES[scene])
        t = evaluate(sh, fx, ctx, g, scene, strict=not args.default_mode)
        print("BF: %g" % t["seconds"])
        with open(os.path.join(args.out, "input_map_0.pgm"), "wb") as f:
            f.write(fx.pgm_bytes(t["input_prob"]))
        with open(os.path.join(args.out, "sss_map_0.pgm"), "wb") as f:
            f.write(fx.pgm_bytes(t["sss_prob"]))
        print("  %d poses x %d beams, %.3g pose-candidates*beams/s; best %.6g at delta (%g, %g, %g)"
              % (t["n_calls"], t["beams"], t["n_calls"] * t["beams"] / t["seconds"], t["prob"], *t["delta"]),
              file=sys.stderr)
    return 0


if __name__ == "__main__":
    sys.exit(main())
