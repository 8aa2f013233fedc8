import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import numpy as np
import __graft_entry__ as ge
from synth import CELL_OCC, CELL_TBM, make_scene
pkg = ge.load_package()
ctx = pkg.Context(0)
STRICT = dict(sum_order=1, pose_trig=1)
modes = {"seq+devtrig": dict(sum_order=1, pose_trig=0), "tree+hosttrig": dict(sum_order=0, pose_trig=1),
         "tree+devtrig(chain)": dict()}
div = {k: 0 for k in modes}
first = {k: [] for k in modes}
matches = 0
for seed in range(40):
    cell = CELL_TBM if seed % 3 == 0 else CELL_OCC
    sc = make_scene(cell_model=cell, size=500, scale=0.05, n_beams=360 + 90 * (seed % 5), seed=100 + seed,
                    weighting="viny" if cell == CELL_TBM else "even")
    ctx.upload_map(0, sc["map"])
    c, s = pkg.beam_trig(sc["scan"].angle)
    ctx.scan_upload(sc["scan"].range, c, s, sc["scan"].weight, sc["scan"].factor)
    rs = np.random.RandomState(seed)
    prm = [6 + 7 * (seed % 4), 0.1, 0.1]
    ms = {k: pkg.Matcher(ctx, "HC", pkg.spe_cfg(**v), prm) for k, v in modes.items()}
    strict = pkg.Matcher(ctx, "HC", pkg.spe_cfg(**STRICT), prm)
    for rep in range(5):
        init = sc["true_pose"] + rs.randn(3) * [0.08, 0.08, 0.04]
        b = strict.process_scan(0, init, trace=True)
        matches += 1
        for k, m in ms.items():
            a = m.process_scan(0, init, trace=True)
            same = (a["n_calls"] == b["n_calls"] and np.array_equal(a["accepted"], b["accepted"])
                    and np.array_equal(a["poses"], b["poses"]))
            if not same:
                div[k] += 1
                n = min(a["n_calls"], b["n_calls"])
                bad = np.nonzero((a["accepted"][:n] != b["accepted"][:n]) | (a["poses"][:n] != b["poses"][:n]).any(1))[0]
                i = int(bad[0]) if len(bad) else n
                # running best before call i in strict trace
                acc = np.nonzero(b["accepted"][:i])[0]
                best = b["scores"][acc[-1]] if len(acc) else float("nan")
                first[k].append((seed, rep, i, b["n_calls"], best, b["scores"][i] if i < n else None, a["scores"][i] if i < n else None,
                                 cell))
print("matches", matches, "divergences", div)
for k, v in first.items():
    for e in v:
        print(k, e, "rel diff strict cand-best: %.3g" % ((e[5] - e[4]) / e[4]) if e[5] is not None else "")
