"""Where the microseconds of the raw-scan step go (bench.py's headline step = slamhip_scan_filter_upload +
slamhip_matcher_process_scan): host time of the upload call alone, the match behind an upload, the match on a scan
already in HBM.  Run on the GPU box."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402
import bench  # noqa: E402
from synth import make_scene  # noqa: E402

pkg = ge.load_package()
ctx = pkg.Context(0)
sc = make_scene(cell_model=0, size=2000, scale=0.05, n_beams=1080, seed=100)
ctx.upload_map(0, sc["map"])
scenes = bench.rotating_scenes(sc, 1080, "even")
m = pkg.Matcher(ctx, "HC", pkg.spe_cfg(), [128, 0.1, 0.1])
ups = [ctx.make_raw_scan(0, s["raw_range"], s["raw_angle"], is_occ=s["is_occ"]) for s in scenes]
for j, s in enumerate(scenes):
    c_, s_ = pkg.beam_trig(s["angle"])
    ctx.scan_store(j, s["range"], c_, s_, s["weight"])
N = 400


def timed(fn):
    for i in range(32):
        fn(i % 16)
    ctx.synchronize()
    t0 = time.perf_counter()
    for i in range(N):
        fn(i % 16)
    ctx.synchronize()
    return 1e6 * (time.perf_counter() - t0) / N


def f_upload(k):
    ups[k](scenes[k]["init_pose"])


def f_upload_sync(k):
    ups[k](scenes[k]["init_pose"])
    ctx.synchronize()


def f_raw(k):
    ups[k](scenes[k]["init_pose"])
    m.process_scan(0, scenes[k]["init_pose"])


def f_res(k):
    ctx.scan_select(k)
    m.process_scan(0, scenes[k]["init_pose"])


print("upload call alone (host side, copies queued): %.1f us" % timed(f_upload))
print("upload call + wait for the copy:              %.1f us" % timed(f_upload_sync))
print("raw step (upload + match):                    %.1f us" % timed(f_raw))
print("resident step (select + match):               %.1f us" % timed(f_res))
