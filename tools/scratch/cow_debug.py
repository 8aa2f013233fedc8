import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/oracle")
import numpy as np
import __graft_entry__ as ge
from helpers import load
pkg = ge.load_package()
g = load("particle_maps_cow.npz")
w, h = [int(v) for v in g["size"]]
scale, blur, shift = float(g["scale"]), float(g["blur"]), float(g["shift_amount"])
ox, oy = [int(v) for v in g["origin"]]
base = tuple(g["base"])
ctx = pkg.Context(0)
def dense(mid):
    ctx.map_bind(mid, 2, w, h, g["origin"], scale, g["unknown"][:3])
    c0, s0 = pkg.beam_trig(g["scan0_angle"])
    ctx.map_append_scan(mid, pkg.RULE_GMAPPING, g["pose0"], g["scan0_range"], c0, s0, is_occ=g["scan0_occ"], base=base,
                        blur=blur, estimator=1, shift_amount=shift)
dense(4)
a = ctx.map_download_aux(4, 0, 0, w, h, 2)
print("ancestor dense aux equal:", np.array_equal(a, g["A_aux"]))
# single-scan path for every B pose
c1, s1 = pkg.beam_trig(g["scan1_angle"])
for i in range(6):
    dense(5)
    ctx.map_append_scan(5, pkg.RULE_GMAPPING, g["poses_b"][i], g["scan1_range"], c1, s1, is_occ=g["scan1_occ"], base=base,
                        blur=blur, estimator=1, shift_amount=shift)
    a = ctx.map_download_aux(5, 0, 0, w, h, 2)
    bad = np.argwhere(a != g["B%d_aux" % i])
    print("single-scan B%d mismatches:" % i, len(bad), bad[:3].tolist(), [ (a[tuple(b)], g["B%d_aux" % i][tuple(b)]) for b in bad[:3]])
n = 8
pf = pkg.GmappingFilter(ctx, pkg.gmapping_params(), n, np.arange(n, dtype=np.uint32))
pf.enable_particle_maps(4, extent_tiles=8, pool_tiles=16 + 24 * n, base=base, blur=blur, estimator=1, shift_amount=shift)
pf.particle_maps_append(np.arange(6), g["poses_b"], g["scan1_range"], g["scan1_angle"], g["scan1_occ"])
for i in range(6):
    p, a = pf.particle_map(i, -ox, -oy, w, h)
    bad = np.argwhere(a != g["B%d_aux" % i])
    print("batch B%d mismatches:" % i, len(bad), bad[:3].tolist(), [(a[tuple(b)], g["B%d_aux" % i][tuple(b)]) for b in bad[:3]])
    # which beam ends there?
    for b in bad[:1]:
        cy, cx = b[0] - oy, b[1] - ox
        ps = g["poses_b"][i]
        ang = g["scan1_angle"]; r = g["scan1_range"]
        ex = ps[0] + r * np.cos(ps[2] + ang); ey = ps[1] + r * np.sin(ps[2] + ang)
        ecx = np.floor(ex / scale); ecy = np.floor(ey / scale)
        hit = np.nonzero((ecx == cx) & (ecy == cy))[0]
        print("   cell", cx, cy, "beams ending there", hit.tolist(), "frac", [(ex[k]/scale - ecx[k], ey[k]/scale-ecy[k]) for k in hit], "occ", g["scan1_occ"][hit].tolist())
