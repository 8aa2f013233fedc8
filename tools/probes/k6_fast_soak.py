"""scratch: the batched map update's free-space fast path against the sorted pipeline over many random batches
(scenes, pose clouds, adders) -- every particle's whole map byte for byte.  python tools/scratch/k6_fast_soak.py [n]"""
import os, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import __graft_entry__ as ge
from synth import make_scene
pkg = ge.load_package()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0
for seed in range(rounds):
    rs = np.random.RandomState(1000 + seed)
    size = int(rs.choice([384, 512, 768]))
    scale = float(rs.choice([0.05, 0.1, 0.025]))
    beams = int(rs.choice([180, 360, 1080]))
    n = int(rs.randint(2, 24))
    sc = make_scene(cell_model=2, size=size, scale=scale, n_beams=beams, seed=seed)
    m, scan = sc["map"], sc["scan"]
    spread = rs.choice([0.01, 0.1, 0.5])
    poses = sc["true_pose"] + rs.randn(n, 3) * [spread, spread, 0.05]
    adder = [{}, {"blur": 0.2}, {"blur": 0.3, "estimator": 1, "shift_amount": 0.01 * scale},
             {"estimator": 1, "shift_amount": 0.01 * scale}, {"blur": -0.03, "max_range": float(np.percentile(scan.range, 80))}][seed % 5]
    occ = None if seed % 3 else (rs.rand(scan.n) < 0.8).astype(np.int32)
    out = {}
    for mode in ("1", "0"):
        ctx = pkg.Context(0)
        ctx.set_option(pkg.OPT_K6_BATCH_FAST, int(mode))
        ctx.map_bind(3, 2, size, size, m.origin, scale, m.unknown)
        if seed % 4:
            ctx.map_upload_window(3, 0, 0, m.payload)
        pf = pkg.GmappingFilter(ctx, pkg.gmapping_params(), n, np.arange(n, dtype=np.uint32))
        pf.enable_particle_maps(3, extent_tiles=(size + 127) // 128 + 2, pool_tiles=64 + 40 * n, **adder)
        rs2 = np.random.RandomState(seed)
        tot = [pf.particle_maps_append(np.arange(n), poses + 0.02 * k * rs2.randn(1, 3), scan.range, scan.angle, occ) for k in range(3)]
        ox, oy = m.origin
        out[mode] = (tot, [tuple(a.tobytes() for a in pf.particle_map(i, -ox, -oy, size, size)) for i in range(n)])
        pf.close(); ctx.close()
    ok = out["1"] == out["0"]
    bad += 0 if ok else 1
    print("round %d: %d particles, %d beams, size %d @ %.3f, adder %s -> %s (%d updates)" % (seed, n, beams, size, scale, adder, "equal" if ok else "DIFFERENT", sum(out["1"][0])))
print("soak: %d rounds, %d different" % (rounds, bad))
sys.exit(1 if bad else 0)
