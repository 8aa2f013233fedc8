"""Synthetic worlds, scans and SLAM maps at BASELINE.json sizes (numpy only).

The reference's own generators (src/utils/data_generation/*) cannot travel to the GPU box, so the
bench and the full-size property tests build their inputs here: a rooms+corridors ground-truth
raster, a ray-cast 270-degree scan with N(0, 0.01 m) range noise from a fixed seed, and a SLAM map
made by inserting several scans from jittered poses so that cells hold running-mean / TBM /
GMapping values rather than binary ones (SURVEY 8d "Concrete synthetic inputs").  The numbers need
not equal the reference generator's -- parity is always checked between the HIP path and the
oracle ON THE SAME inputs; the golden fixtures cover the reference's own generator.

This module imports neither the oracle nor the product package.
"""
import numpy as np

CELL_OCC, CELL_TBM, CELL_GMAPPING = 0, 1, 2
STRIDE = {CELL_OCC: 1, CELL_TBM: 4, CELL_GMAPPING: 3}


class MapData:
    def __init__(self, cell_model, payload, origin, scale, unknown, bounded=False):
        self.cell_model = int(cell_model)
        self.payload = np.ascontiguousarray(payload, dtype=np.float64)
        self.height, self.width = self.payload.shape[:2]
        self.origin = (int(origin[0]), int(origin[1]))
        self.scale = float(scale)
        u = np.zeros(4)
        u[:STRIDE[self.cell_model]] = np.asarray(unknown, dtype=np.float64).ravel()[:STRIDE[self.cell_model]]
        self.unknown = u
        self.bounded = bool(bounded)


class Scan:
    def __init__(self, rng, ang, weight, factor=None):
        self.range = np.ascontiguousarray(rng, dtype=np.float64)
        self.angle = np.ascontiguousarray(ang, dtype=np.float64)
        self.weight = np.ascontiguousarray(weight, dtype=np.float64)
        self.factor = (np.ones(self.range.size) if factor is None
                       else np.ascontiguousarray(factor, dtype=np.float64))
        self.trig_mode = 0

    @property
    def n(self):
        return int(self.range.size)


def make_world(size, scale, seed=0):
    """Ground-truth occupancy raster [size, size] (bool), rooms + corridors, robot near the centre."""
    rs = np.random.RandomState(seed)
    gt = np.zeros((size, size), dtype=bool)
    span_m = size * scale
    half = int(min(0.45 * size, 28.0 / scale))  # outer box, at most ~56 m across
    c = size // 2
    lo, hi = c - half, c + half
    gt[lo, lo:hi + 1] = gt[hi, lo:hi + 1] = True
    gt[lo:hi + 1, lo] = gt[lo:hi + 1, hi] = True
    # interior walls with door gaps, kept away from the robot's cell
    room = max(int(6.0 / scale), 8)
    door = max(int(1.2 / scale), 3)
    for k, pos in enumerate(range(lo + room, hi - room // 2, room)):
        if abs(pos - c) < room // 3:
            continue
        gaps = rs.randint(lo + door, hi - 2 * door, size=3)
        line = np.ones(hi - lo + 1, dtype=bool)
        for g in gaps:
            line[g - lo:g - lo + door] = False
        if k % 2 == 0:
            gt[pos, lo:hi + 1] |= line
        else:
            gt[lo:hi + 1, pos] |= line
    # a few pillars
    for _ in range(12):
        px, py = rs.randint(lo + 3, hi - 3, size=2)
        if abs(px - c) < room // 4 and abs(py - c) < room // 4:
            continue
        w = max(int(0.4 / scale), 1)
        gt[py:py + w, px:px + w] = True
    del span_m
    return gt


def cast_scan(gt, scale, pose, n_beams, fov_deg=270.0, max_dist=30.0, noise=0.01, seed=42, raw=False):
    """Ray-cast a scan from `pose` on the raster (origin at the raster centre).  raw: the scan as the scanner hands
    it over -- every beam, (ranges, angles, is_occupied) with max_dist on the beams that hit nothing -- instead of
    the hits alone; ranges[is_occupied] are the same numbers either way."""
    size = gt.shape[0]
    org = size // 2
    ang = np.deg2rad(-fov_deg / 2 + fov_deg / n_beams * np.arange(n_beams))
    step = scale / 4.0
    t = np.arange(step, max_dist, step)
    d = pose[2] + ang
    chunk = 128
    ranges = np.full(n_beams, np.inf)
    for b0 in range(0, n_beams, chunk):
        dd = d[b0:b0 + chunk, None]
        x = pose[0] + t[None, :] * np.cos(dd)
        y = pose[1] + t[None, :] * np.sin(dd)
        ix = np.floor(x / scale).astype(np.int64) + org
        iy = np.floor(y / scale).astype(np.int64) + org
        inb = (ix >= 0) & (ix < size) & (iy >= 0) & (iy < size)
        hit = np.zeros_like(inb)
        hit[inb] = gt[iy[inb], ix[inb]]
        first = hit.argmax(axis=1)
        has = hit.any(axis=1)
        r = np.where(has, t[first] + scale / 2, np.inf)
        ranges[b0:b0 + chunk] = r
    ok = np.isfinite(ranges)
    rs = np.random.RandomState(seed)
    ranges = ranges + rs.randn(n_beams) * noise
    if raw:
        return np.where(ok, ranges, max_dist), ang, ok.astype(np.int32)
    return ranges[ok], ang[ok]


def _insert_counts(size, scale, pose, rng, ang, blur_m):
    """Per-cell (n_free, n_occ, sum of blurred occupied probs weights, obstacle sums) of one scan."""
    org = size // 2
    step = scale / 2.0
    n = rng.size
    kmax = int(np.ceil(rng.max() / step)) + 1
    t = np.arange(kmax)[None, :] * step
    d = (pose[2] + ang)[:, None]
    ex = pose[0] + rng * np.cos(pose[2] + ang)
    ey = pose[1] + rng * np.sin(pose[2] + ang)
    ecx = np.floor(ex / scale).astype(np.int64)
    ecy = np.floor(ey / scale).astype(np.int64)
    valid = t < (rng[:, None] - scale * 0.75)
    x = pose[0] + t * np.cos(d)
    y = pose[1] + t * np.sin(d)
    cx = np.floor(x / scale).astype(np.int64)
    cy = np.floor(y / scale).astype(np.int64)
    not_end = (cx != ecx[:, None]) | (cy != ecy[:, None])
    valid &= not_end
    beam = np.broadcast_to(np.arange(n)[:, None], cx.shape)
    key = (beam[valid] * size + (cy[valid] + org)) * size + (cx[valid] + org)
    key = np.unique(key)
    cell = key % (size * size)
    kb = key // (size * size)
    free = np.bincount(cell, minlength=size * size).astype(np.float64)
    # wall blur: cells within blur of the obstacle get a scaled occupied probability
    blur_sum = np.zeros(size * size)
    blur_cnt = np.zeros(size * size)
    if blur_m > 0:
        hole = blur_m / scale
        fy, fx = (cell // size) - org, (cell % size) - org
        d2 = (fx - ecx[kb]) ** 2 + (fy - ecy[kb]) ** 2
        m = d2 < hole * hole
        np.add.at(blur_sum, cell[m], 0.95 * (1.0 - d2[m] / (hole * hole)))
        np.add.at(blur_cnt, cell[m], 1.0)
    ecell = (ecy + org) * size + (ecx + org)
    inb = (ecx + org >= 0) & (ecx + org < size) & (ecy + org >= 0) & (ecy + org < size)
    occ = np.bincount(ecell[inb], minlength=size * size).astype(np.float64)
    ox = np.bincount(ecell[inb], weights=ex[inb], minlength=size * size)
    oy = np.bincount(ecell[inb], weights=ey[inb], minlength=size * size)
    return free, occ, blur_sum, blur_cnt, ox, oy


def build_map(cell_model, size, scale, true_pose, rng, ang, n_scans=5, blur_m=0.3, quality=0.9,
              seed=1, tbm_quals=(0.04, 0.003)):
    rs = np.random.RandomState(seed)
    N = size * size
    free = np.zeros(N)
    occ = np.zeros(N)
    bsum = np.zeros(N)
    bcnt = np.zeros(N)
    ox = np.zeros(N)
    oy = np.zeros(N)
    for _ in range(n_scans):
        p = np.asarray(true_pose) + rs.randn(3) * [0.01, 0.01, 0.002]
        f, o, bs, bc, sx, sy = _insert_counts(size, scale, p, rng, ang, blur_m)
        free += f
        occ += o
        bsum += bs
        bcnt += bc
        ox += sx
        oy += sy
    org = (size // 2, size // 2)
    if cell_model == CELL_OCC:
        # MeanProbabilityCell: mean of 0.5 + (p - 0.5) * quality over the observations
        p_occ = 0.5 + (0.95 - 0.5) * quality
        p_free = 0.5 + (0.01 - 0.5) * quality
        plain_free = free - bcnt
        tot = free + occ
        val = occ * p_occ + plain_free * p_free + (0.5 * bcnt + (bsum - 0.5 * bcnt) * quality)
        pay = np.where(tot > 0, val / np.maximum(tot, 1), 0.5)
        return MapData(CELL_OCC, pay.reshape(size, size, 1), org, scale, [0.5])
    if cell_model == CELL_TBM:
        # TbmBaseCell: conjunctive combination + conflict normalisation per observation
        qo, qe = tbm_quals[0] * quality, tbm_quals[1] * quality
        u = np.ones(N)
        e = np.zeros(N)
        o = np.zeros(N)

        def combine(mask, eu, ee, eo):
            nu = u[mask] * eu
            ne = e[mask] * ee + e[mask] * eu + u[mask] * ee
            no = o[mask] * eo + o[mask] * eu + u[mask] * eo
            w = nu + ne + no
            w[w == 0] = 1.0
            u[mask], e[mask], o[mask] = nu / w, ne / w, no / w

        occ_obs = (1.0 - 0.95 * qo - 0.05 * qo, 0.05 * qo, 0.95 * qo)
        free_obs = (1.0 - 0.01 * qe - 0.99 * qe, 0.99 * qe, 0.01 * qe)
        for k in range(int(max(occ.max(), 1))):
            m = occ > k
            if not m.any():
                break
            combine(m, *occ_obs)
        for k in range(int(min(free.max(), 60))):
            m = free > k
            if not m.any():
                break
            combine(m, *free_obs)
        pay = np.stack([u, e, o, np.zeros(N)], axis=1)
        return MapData(CELL_TBM, pay.reshape(size, size, 4), org, scale, [1.0, 0.0, 0.0, 0.0])
    # GmappingBaseCell: prob = sum(p_occ over hits) / tries, obst = mean endpoint; unknown = -1
    tries = free + occ
    prob = np.where(tries > 0, occ * 0.95 / np.maximum(tries, 1), -1.0)
    mx = np.where(occ > 0, ox / np.maximum(occ, 1), 0.0)
    my = np.where(occ > 0, oy / np.maximum(occ, 1), 0.0)
    pay = np.stack([prob, mx, my], axis=1)
    return MapData(CELL_GMAPPING, pay.reshape(size, size, 3), org, scale, [-1.0, 0.0, 0.0])


def viny_weights(rng, ang):
    ac = np.abs(np.cos(ang))
    w = np.abs(np.sin(ang)) + ac
    w = np.where(ac > 0.9, 3.0, np.where(ac > 0.8, 2.0, w))
    return w * np.sqrt(rng)


def make_scene(cell_model=CELL_OCC, size=2000, scale=0.05, n_beams=1080, seed=0, weighting="even",
               max_dist=30.0, blur_m=0.3):
    """One BASELINE-style scene: map + filtered scan + perturbed initial pose.
    Scan points whose endpoint (at the initial pose) leaves the window are kept: the map is
    treated as unbounded (has_cell always true), like the reference's Unbounded* maps."""
    gt = make_world(size, scale, seed)
    true_pose = np.array([scale / 2, scale / 2, np.deg2rad(90.0)])
    rng, ang = cast_scan(gt, scale, true_pose, n_beams, max_dist=max_dist)
    m = build_map(cell_model, size, scale, true_pose, rng, ang, blur_m=blur_m if cell_model != CELL_GMAPPING else 0.0)
    w = np.full(rng.size, 1.0 / rng.size) if weighting == "even" else viny_weights(rng, ang)
    init = true_pose + np.array([0.07, -0.04, 0.03])
    return dict(map=m, scan=Scan(rng, ang, w), init_pose=init, true_pose=true_pose, gt=gt)
