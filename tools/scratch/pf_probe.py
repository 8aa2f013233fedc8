import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
from synth import make_scene
pkg = ge.load_package(); ctx = pkg.Context(0)
sc = make_scene(cell_model=2, size=4000, scale=0.05, n_beams=1080, seed=4)
ctx.upload_map(1, sc["map"]); scan = sc["scan"]
gp = [0.0, 0.1, 0.0, 0.03, 0.0, 0.0, 0.0, 0.0]
rs = np.random.RandomState(5)
deltas = [sc["true_pose"]] + [rs.randn(3) * [0.05, 0.05, 0.02] for _ in range(40)]
for n in (100, 50, 25, 13):
    f = pkg.GmappingFilter(ctx, pkg.gmapping_params(gp8=gp), n, np.arange(1000, 1000 + n, dtype=np.uint32))
    ts, ls = [], []
    for k in range(14):
        ctx.synchronize(); t0 = time.perf_counter()
        f.predict_match(1, scan.range, scan.angle, None, deltas[k])
        ctx.synchronize(); ts.append(1e3 * (time.perf_counter() - t0)); ls.append(f.stats()["launches"])
    print(n, "ms per step", " ".join("%.3f" % x for x in ts), "| kernels", ls)
    f.close()
