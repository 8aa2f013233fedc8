import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/oracle")
import numpy as np
import pyoracle as po
from pyoracle_mapupdate import RULE_GMAPPING, append_scan_ex
from helpers import load
g = load("particle_maps_cow.npz")
w, h = [int(v) for v in g["size"]]
scale, blur, shift = float(g["scale"]), float(g["blur"]), float(g["shift_amount"])
ox, oy = [int(v) for v in g["origin"]]
base = tuple(g["base"])
O = po.Oracle()
R = po.Ref()
pin = R.map_create(po.REF_CELL_MEAN, po.MAP_UNBOUNDED_PLAIN, 20, 20, scale)
R.append_scan(pin, R.scan_create([0.005], [0.0]), (scale / 2, scale / 2, 0.0), occ_est=1)
i, beam = 0, 237
pose = g["poses_b"][i]
r = g["scan1_range"][beam:beam + 1]; a = g["scan1_angle"][beam:beam + 1]; o = g["scan1_occ"][beam:beam + 1]
m = po.GridMapData(po.CELL_GMAPPING, np.tile(g["unknown"][:3], (h, w, 1)).astype(np.float64), g["origin"], scale, g["unknown"][:3])
aux = np.zeros((h, w, 2))
n = append_scan_ex(O, m, aux, RULE_GMAPPING, pose, r, a, o, base=base, blur=0.0, est_kind=1, shift_amount=shift)
print("oracle updates", n, "end cell payload", m.payload[276, 235], "aux", aux[276, 235], "unknown", g["unknown"])
rm = R.map_create(po.REF_CELL_GMAPPING, po.MAP_UNBOUNDED_LAZY_TILED, w, h, scale, 0.5)
R.append_scan(rm, R.scan_create(r, a, o), pose, occ_est=1, base=base, blur=0.0)
print("reference end cell payload", rm.to_data().payload[276, 235], "aux", rm.aux()[276, 235])
nz = np.argwhere(aux[..., 1] > 0)
print("cells touched", len(nz), "last few", nz[-3:].tolist(), "hits cells", np.argwhere(aux[..., 0] > 0).tolist())
