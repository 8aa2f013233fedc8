"""Collect hill-climbing outcome sequences (per round: 0 = all rejected, j = candidate j-1 accepted last)
from the CPU oracle on bench-like scenes; save for shape experiments."""
import sys, os, pickle
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/oracle")
import numpy as np
import pyoracle as po
from synth import make_scene
O = po.Oracle()
seqs = []
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    sc = make_scene(cell_model=0, size=1000, scale=0.05, n_beams=720, seed=100 + seed)
    rs = np.random.RandomState(seed)
    for rep in range(6):
        init = sc["true_pose"] + rs.randn(3) * [0.08, 0.08, 0.04] if rep else sc["init_pose"]
        for prm in ([128, 0.1, 0.1], [6, 0.1, 0.1]):
            e = O.enumerator(po.SM_HC, prm)
            r = O.process_scan(e, sc["map"], sc["scan"], po.make_cfg(), init)
            acc = np.asarray(r["accepted"])[1:]  # drop the initial pose
            n_full = len(acc) // 6
            outs = []
            for k in range(n_full):
                a = acc[6 * k:6 * k + 6]
                nz = np.nonzero(a)[0]
                outs.append(int(nz[-1]) + 1 if len(nz) else 0)
            seqs.append((prm[0], outs, len(acc) - 6 * n_full))
pickle.dump(seqs, open("/tmp/hc_seqs.pkl", "wb"))
print(len(seqs), "matches", sum(len(s[1]) for s in seqs), "rounds")
for s in seqs[:4]:
    print(s[0], "".join(str(o) for o in s[1]), s[2])
