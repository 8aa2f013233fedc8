import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import __graft_entry__ as ge
import pyoracle as po
from synth import make_scene, CELL_GMAPPING
pkg = ge.load_package(); ctx = pkg.Context(0); O = po.Oracle()
shown = 0
for seed in range(12):
    sc = make_scene(cell_model=CELL_GMAPPING, size=500, scale=0.05, n_beams=360 + 90 * (seed % 5), seed=500 + seed)
    ctx.upload_map(0, sc["map"]); c, s = pkg.beam_trig(sc["scan"].angle)
    ctx.scan_upload(sc["scan"].range, c, s, sc["scan"].weight, sc["scan"].factor)
    rs = np.random.RandomState(2000 + seed); prm = [6 + 7 * (seed % 4), 0.1, 0.1]
    m = pkg.Matcher(ctx, "HC", pkg.spe_cfg(oope=pkg.OOPE_GMAPPING), prm); m.set_device_chain(0)
    mh = pkg.Matcher(ctx, "HC", pkg.spe_cfg(oope=pkg.OOPE_GMAPPING, pose_trig=1), prm); mh.set_device_chain(0)
    for rep in range(5):
        init = sc["true_pose"] + rs.randn(3) * [0.08, 0.08, 0.04]
        b = O.process_scan(O.enumerator(po.SM_HC, prm), sc["map"], sc["scan"], po.make_cfg(oope=po.OOPE_GMAPPING), init, cache=po.Oracle.new_gm_cache())
        ctx.gm_cache_reset(); a = m.process_scan(0, init, trace=True)
        ctx.gm_cache_reset(); h = mh.process_scan(0, init, trace=True)
        for name, x in (("dev-trig", a), ("host-trig", h)):
            n = min(x["n_calls"], b["n_calls"])
            bad = np.nonzero(x["accepted"][:n] != b["accepted"][:n])[0]
            if len(bad) or x["n_calls"] != b["n_calls"]:
                i = int(bad[0]) if len(bad) else n
                acc = np.nonzero(b["accepted"][:i])[0]; best = b["scores"][acc[-1]]
                acx = np.nonzero(x["accepted"][:i])[0]; bestx = x["scores"][acx[-1]]
                print(seed, rep, name, "calls", x["n_calls"], b["n_calls"], "first bad", i, "oracle: cand %.17g best %.17g acc %d | hip: cand %.17g best %.17g acc %d | pose equal %s"
                      % (b["scores"][i], best, b["accepted"][i], x["scores"][i], bestx, x["accepted"][i], np.array_equal(x["poses"][i], b["poses"][i])))
                shown += 1
print("shown", shown)
