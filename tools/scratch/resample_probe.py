import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
from synth import make_scene
pkg = ge.load_package(); ctx = pkg.Context(0)
sc = make_scene(cell_model=2, size=4000, scale=0.05, n_beams=1080, seed=4)
ctx.upload_map(1, sc["map"]); scan = sc["scan"]
gp = [0.0, 0.1, 0.0, 0.03, 0.0, 0.0, 0.0, 0.0]
big = [np.array(v) for v in ([0.4, 0.5, 0.3], [0.5, -0.4, 0.25], [-0.45, 0.5, -0.3], [-0.5, -0.45, -0.25])]
def pattern(name, n_steps):
    rs = np.random.RandomState(9); out = [sc["true_pose"]]
    for k in range(n_steps):
        if name == "test":
            d = [[0.02, 0.01, 0.01], [0.4, 0.5, 0.3], [0.01, -0.02, 0.02], [0.5, -0.4, 0.25], [0.02, 0.02, 0.0], [0.45, 0.5, -0.3], [0.0, 0.01, 0.01]]
            out.append(np.array(d[k % 7]))
        elif name == "cycle3":
            out.append(big[(k // 3) % 4] if k % 3 == 1 else rs.randn(3) * [0.05, 0.05, 0.02])
        elif name == "cycle2":
            out.append(big[(k // 2) % 4] if k % 2 == 1 else rs.randn(3) * [0.05, 0.05, 0.02])
    return out
for name in ("test", "cycle3", "cycle2"):
    for n in (100, 13):
        f = pkg.GmappingFilter(ctx, pkg.gmapping_params(gp8=gp), n, np.arange(1000, 1000 + n, dtype=np.uint32))
        dl = pattern(name, 14); rsm = []
        for k in range(13):
            rq, _ = f.step(1, scan.range, scan.angle, None, dl[k], 7 + k)
            rsm.append(int(rq))
        print(name, n, rsm)
        f.close()
