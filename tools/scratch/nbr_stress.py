"""Stress of the neighbourhood masks (tests/test_gpu_nbr_masks.py at length): hundreds of random map writes of every
kind on one dense GMAPPING window, masks checked after each, scores against a fresh upload every so often."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402
from synth import make_scene  # noqa: E402

pkg = ge.load_package()
ctx, fresh = pkg.Context(0, testing=True), pkg.Context(0, testing=True)
sc = make_scene(cell_model=2, size=800, scale=0.05, n_beams=1080, seed=31)
m, scan = sc["map"], sc["scan"]
c, s = pkg.beam_trig(scan.angle)
cfg = pkg.spe_cfg(oope=pkg.OOPE_GMAPPING)
r = np.random.default_rng(77)
poses = np.tile(sc["true_pose"], (64, 1)) + r.uniform(-0.4, 0.4, (64, 3)) * [1, 1, 0.3]


def masks():
    v, b = C.c_int(-1), C.c_longlong(-1)
    assert ctx.L.slamhip_map_debug_nbr_masks(ctx.h, 0, C.byref(v), C.byref(b)) == 0
    return v.value, b.value


def score(cx, mid):
    cx.gm_cache_reset()
    cx.scan_upload(scan.range, c, s, scan.weight)
    return cx.score_poses(mid, cfg, poses)


ctx.upload_map(0, m)
score(ctx, 0)
assert masks() == (1, 0)
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for it in range(n_iter):
    kind = r.integers(0, 10)
    if kind < 7:  # K6, one of the three paths
        ctx.set_option(pkg.OPT_K6_PATH, int(r.integers(0, 3)))
        pose = sc["true_pose"] + np.array([r.uniform(-3, 3), r.uniform(-3, 3), r.uniform(-3.1, 3.1)])
        rng = np.minimum(scan.range, 15.0) * r.uniform(0.2, 1.0, scan.range.size if r.integers(0, 2) else 1)
        ctx.map_append_scan(0, pkg.RULE_GMAPPING, pose, rng, c, s, None)
    elif kind < 9:  # dirty log
        n = int(r.integers(1, 2000))
        xy = r.integers(0, m.width, (n, 2))
        vals = np.zeros((n, 3))
        vals[:, 0] = r.choice([0.0, 0.05, 0.1, 0.3, 0.9, -1.0], n)
        vals[:, 1:] = r.uniform(-20, 20, (n, 2))
        ctx.map_apply_dirty(0, xy, vals)
    else:  # partial upload
        w, h = int(r.integers(1, 200)), int(r.integers(1, 200))
        x0, y0 = int(r.integers(0, m.width - w + 1)), int(r.integers(0, m.height - h + 1))
        patch = np.zeros((h, w, 3))
        patch[..., 0] = r.choice([0.0, 0.09, 0.1, 0.7], (h, w))
        patch[..., 1:] = r.uniform(-20, 20, (h, w, 2))
        ctx.map_upload_window(0, x0, y0, patch)
    v = masks()
    assert v == (1, 0), (it, kind, v)
    if it % 25 == 24:
        cells = ctx.map_download_window(0, 0, 0, m.width, m.height, 3)
        fresh.map_bind(1, pkg.CELL_GMAPPING, m.width, m.height, m.origin, m.scale, m.unknown)
        fresh.map_upload_window(1, 0, 0, cells)
        a, b = score(ctx, 0), score(fresh, 1)
        fresh.map_release(1)
        assert np.array_equal(a, b), it
print("nbr stress: %d writes, masks and scores equal throughout" % n_iter)
