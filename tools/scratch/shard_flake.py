import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
import test_gpu_shard as T
from synth import make_scene
pkg = ge.load_package()
sc = make_scene(cell_model=2, size=800, scale=0.05, n_beams=360, seed=4)
n, world = 21, 3
steps = T.deltas(sc, 8)
ctx = pkg.Context(0); ctx.upload_map(1, sc["map"])
ref = []
whole = pkg.GmappingFilter(ctx, pkg.gmapping_params(gp8=T.GP), n, np.arange(2000, 2000 + n, dtype=np.uint32))
for k, d in enumerate(steps):
    rw, iw = whole.step(1, sc["scan"].range, sc["scan"].angle, None, d, 7 + k)
    ref.append((rw, iw.copy(), *whole.state()))
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    logs, counts = T.run_loopback_ranks(pkg, world, n, sc, steps, sc["scan"], T.GP, "flake-%d" % it)
    for k in range(len(steps)):
        pw = ref[k][2]
        got = np.concatenate([logs[r][0][k][2] for r in range(world)])
        if not np.array_equal(got, pw):
            first = 0
            for r in range(world):
                pr = logs[r][0][k][2]
                ok = np.array_equal(pr, pw[first:first + counts[r]])
                if not ok:
                    wr = logs[r][0][k][3]
                    print("iter %d step %d rank %d differs: reruns %s, resampled %s (ref %s); first pose got %s want %s; weights equal %s"
                          % (it, k, r, [logs[r][0][j][5] for j in range(len(steps))], logs[r][0][k][0], ref[k][0], pr[0], pw[first],
                             np.array_equal(wr, ref[k][3][first:first + counts[r]])))
                first += counts[r]
            bad += 1
            break
print("bad iterations", bad)
