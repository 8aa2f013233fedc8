import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
from bench_legs.common import WORKLOADS, rotating_scenes
from synth import make_scene
pkg = ge.load_package()
cell, weighting, kind, params, _, _ = WORKLOADS["hc"]
sc = make_scene(cell_model=cell, size=2000, scale=0.05, n_beams=1080, seed=100, weighting=weighting)
scenes = rotating_scenes(sc, 1080, weighting)
ctx = pkg.Context(0)
ctx.upload_map(0, sc["map"])
for j, s_ in enumerate(scenes):
    c_, s__ = pkg.beam_trig(s_["angle"])
    ctx.scan_store(j, s_["range"], c_, s__, s_["weight"])
for K in (9, 10, 11, 12, 16, 20, 22, 24, 28, 32, 40, 48, 64):
    m = pkg.Matcher(ctx, kind, pkg.spe_cfg(), params)
    m.set_device_chain(2)
    blk = m.make_batch([dict(map_id=0, scan_slot=k % 16, init_pose=scenes[k % 16]["init_pose"]) for k in range(K)])
    m.process_scan_batch(blk)
    print(K, m.resident_stats(), m.stats()["kernels_launched"], m.stats()["launches"])
    m.close()
