import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/oracle")
import numpy as np
import __graft_entry__ as ge
import pyoracle as po
from pyoracle_mapupdate import RULE_GMAPPING, append_scan_ex
from helpers import load
pkg = ge.load_package()
g = load("particle_maps_cow.npz")
w, h = [int(v) for v in g["size"]]
scale, blur, shift = float(g["scale"]), float(g["blur"]), float(g["shift_amount"])
base = tuple(g["base"])
ctx = pkg.Context(0)
O = po.Oracle()
for (i, beam) in ((0, 237), (1, 233), (4, 237)):
    for bl in (0.0, blur):
        pose = g["poses_b"][i]
        r = g["scan1_range"][beam:beam + 1]; a = g["scan1_angle"][beam:beam + 1]; o = g["scan1_occ"][beam:beam + 1]
        ctx.map_bind(6, 2, w, h, g["origin"], scale, g["unknown"][:3])
        c, s = pkg.beam_trig(a)
        ctx.map_append_scan(6, pkg.RULE_GMAPPING, pose, r, c, s, is_occ=o, base=base, blur=bl, estimator=1, shift_amount=shift)
        ga = ctx.map_download_aux(6, 0, 0, w, h, 2)
        gp = ctx.map_download_window(6, 0, 0, w, h, 3)
        ctx.map_release(6)
        m = po.GridMapData(po.CELL_GMAPPING, np.tile(g["unknown"][:3], (h, w, 1)).astype(np.float64), g["origin"], scale, g["unknown"][:3])
        aux = np.zeros((h, w, 2))
        append_scan_ex(O, m, aux, RULE_GMAPPING, pose, r, a, o, base=base, blur=bl, est_kind=1, shift_amount=shift)
        bad = np.argwhere(ga != aux)
        print("pose", i, "beam", beam, "blur", bl, "aux mismatches", bad.tolist(), [(ga[tuple(b)], aux[tuple(b)]) for b in bad])
        for b in bad[:2]:
            print("    gpu payload", gp[b[0], b[1]], "oracle payload", m.payload[b[0], b[1]])
        ex = pose[0] + r * np.cos(pose[2] + a); ey = pose[1] + r * np.sin(pose[2] + a)
        print("    endpoint", repr(float(ex[0])), repr(float(ey[0])), "cell frac", ex[0] / scale % 1, ey[0] / scale % 1, "range", repr(float(r[0])), "angle", repr(float(a[0])), "pose", [repr(float(v)) for v in pose])
