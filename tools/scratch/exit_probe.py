import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import __graft_entry__ as ge
from synth import make_scene
mode = sys.argv[1]
pkg = ge.load_package()
ctx = pkg.Context(0)
sc = make_scene(cell_model=0, size=400, scale=0.1, n_beams=360, seed=3)
ctx.upload_map(0, sc["map"])
c, s = pkg.beam_trig(sc["scan"].angle)
ctx.scan_upload(sc["scan"].range, c, s, sc["scan"].weight, sc["scan"].factor)
if mode in ("chain", "chain_close", "chain_ctxfirst"):
    m = pkg.Matcher(ctx, "HC", pkg.spe_cfg(), [6, 0.1, 0.1])
    m.process_scan(0, sc["init_pose"])
    if mode == "chain_close":
        del m
        ctx.close()
    if mode == "chain_ctxfirst":
        ctx.close()
        del m
elif mode == "host":
    m = pkg.Matcher(ctx, "HC", pkg.spe_cfg(), [6, 0.1, 0.1])
    m.set_device_chain(0)
    m.process_scan(0, sc["init_pose"])
elif mode == "rccl":
    ctx.shard_init(0, 1, pkg.shard_unique_id())
elif mode == "rccl_close":
    ctx.shard_init(0, 1, pkg.shard_unique_id())
    ctx.shard_destroy()
    ctx.close()
elif mode == "none":
    pass
print("done", mode)
