"""GPU suite: K6 map update on crafted scans -- the corners of the walk and of the chain kernels that a
random scene seldom visits -- against the oracle, all five cell kinds, bit-exact:
  * poses on cell corners / centres with beams along the axes and diagonals and endpoints on grid lines
    (the fuzzy tie rule of world_to_cells, regular_squares_grid.h:74-98; diagonal steps; the Bresenham
    fail-over when rounding sends the walk astray);
  * a bundle of identical beams (every cell of the bundle is a chain of several hundred records:
    k_mu_apply_long, chains that cross wave boundaries, GMapping free runs interrupted by hits);
  * zero-length beams, beams beyond max_range, empty observations, negative (range-proportional) blur;
  * beams whose walk goes astray and is redone by DiscreteSegment2D (found by a seeded search with a plain
    Python restatement of the walk: end points a hair off grid corners)."""
import math

import numpy as np
import pytest

import __graft_entry__ as ge

pytestmark = pytest.mark.gpu

# name -> (cell model, rule, host stride, aux doubles)
KINDS = {"last": (0, 0, 1, 0), "affine": (0, 1, 1, 0), "mean": (0, 2, 1, 1), "tbm": (1, 3, 4, 0), "gmapping": (2, 4, 3, 2)}
SCALE = 0.1
SIZE = 400


@pytest.fixture(scope="module")
def pkg():
    return ge.load_package()


@pytest.fixture(scope="module")
def ctx(pkg):
    c = pkg.Context(0)
    yield c
    c.close()


K6_PATHS = {"gather": 0, "counting": 1, "radix": 2}


def set_k6_path(pkg, ctx, name):
    ctx.set_option(pkg.OPT_K6_PATH, K6_PATHS[name])


@pytest.fixture(autouse=True)
def _default_k6_path(pkg, ctx):
    set_k6_path(pkg, ctx, "gather")
    yield


def crafted_scans():
    """(pose, range, angle, is_occ, blur, max_range) tuples."""
    out = []
    dirs = np.deg2rad(np.arange(0, 360, 45.0))
    r2 = np.sqrt(2.0)
    # 1. corner / centre poses, axis and diagonal beams, endpoints on grid lines and cell centres
    for px, py in [(0.0, 0.0), (SCALE / 2, SCALE / 2), (3 * SCALE, -2 * SCALE), (0.05, 0.0), (1.0, 1.0)]:
        ang, rng = [], []
        for a in dirs:
            diag = abs(np.sin(2 * a)) > 0.5
            for cells in (0, 1, 7, 30, 77):
                ang.append(a)
                rng.append(cells * SCALE * (r2 if diag else 1.0))
        out.append((np.array([px, py, 0.0]), np.array(rng), np.array(ang), np.ones(len(rng), np.int32), 0.0, np.inf))
        out.append((np.array([px, py, np.deg2rad(45.0)]), np.array(rng), np.array(ang),
                    (np.arange(len(rng)) % 3 != 0).astype(np.int32), 0.25, np.inf))
    # 2. bundles of identical beams: chains of 200 / 333 records on every cell of the bundle, free and
    #    occupied observations mixed, blur on; a second bundle ends INSIDE the first one's free run
    for n, a_deg, rr in [(200, 30.0, 9.0), (333, 200.0, 12.3)]:
        ang = np.full(n, np.deg2rad(a_deg))
        rng = np.full(n, rr)
        rng[n // 3::7] = rr * 0.5          # hits on cells the other beams pass through
        occ = (np.arange(n) % 5 != 0).astype(np.int32)
        out.append((np.array([0.31, -0.17, 0.2]), rng, ang, occ, 0.3, np.inf))
    # 3. a dense fan (every beam crosses the robot's cell and its neighbours: long and short chains mixed)
    ang = np.linspace(-np.pi, np.pi, 720, endpoint=False)
    rs = np.random.RandomState(5)
    rng = 2.0 + 14.0 * rs.rand(720)
    out.append((np.array([-1.234, 2.345, 1.0]), rng, ang, (rs.rand(720) < 0.8).astype(np.int32), 0.2, np.inf))
    # 4. zero-length beams, beams beyond max_range (dropped), negative blur = proportional to range^2
    rng = np.array([0.0, 0.0, 5.0, 25.0, 1e-9, 7.5, 19.99, 20.01])
    ang = np.deg2rad(np.array([0.0, 90.0, 10.0, 20.0, 30.0, 40.0, 50.0, 60.0]))
    out.append((np.array([0.2, 0.2, 0.0]), rng, ang, np.ones(8, np.int32), -0.004, 20.0))
    # 5. rays from a cell centre along (p, q) with p, q odd pass a grid VERTEX every (p, q) cells: 1, 2, 6 and 13 ties
    #    per beam with plain digital-line stretches between them -- the reference steps diagonally and resets its error
    #    term there (regular_squares_grid.h:85-91); the wave walks such a beam piece by piece (mu_walk_beam_wave), the
    #    13-tie ones go to the step-by-step walk
    ang, rng = [], []
    for p_, q_ in [(3, 1), (1, 3), (5, 3), (3, 5), (7, 1), (-3, 1), (3, -1), (-5, -3), (-7, 5), (9, 7), (-1, -9)]:
        for t_end in (0.75, 2.2, 6.3, 12.7):
            if t_end * max(abs(p_), abs(q_)) > 100:
                continue
            ang.append(np.arctan2(q_, p_))
            rng.append(t_end * np.hypot(p_, q_) * SCALE)
    out.append((np.array([SCALE / 2, SCALE / 2, 0.0]), np.array(rng), np.array(ang), np.ones(len(rng), np.int32), 0.0, np.inf))
    out.append((np.array([-3.5 * SCALE, 6.5 * SCALE, 0.0]), np.array(rng), np.array(ang),
                (np.arange(len(rng)) % 4 != 0).astype(np.int32), 0.2, np.inf))
    return out


def _walk_goes_astray(scale, x0, y0, x1, y1):
    """world_to_cells (regular_squares_grid.h:56-101) up to its decision to fall back to Bresenham."""
    def eq(a, b):
        return abs(a - b) <= 1e-7 * max(1.0, abs(a), abs(b))
    dx, dy = x1 - x0, y1 - y0
    ix, iy = (1 if 0 < dx else -1), (1 if 0 < dy else -1)
    px, py = math.floor(x0 / scale), math.floor(y0 / scale)
    ex, ey = math.floor(x1 / scale), math.floor(y1 / scale)
    cells = abs(ex - px) + abs(ey - py) + 1
    e = (dx * y0 + ((px + 0.5) * scale - x0) * dy) - (py + 0.5) * scale * dx
    exi, eyi = ix * scale * dy, -iy * scale * dx
    n = 0
    while True:
        n += 1
        if px == ex and py == ey:
            return False
        if cells < n:
            return True
        e_x, e_y = e + exi, e + eyi
        d = abs(e_y) - abs(e_x)
        if eq(d, 0):
            if px == ex:
                py += iy
            elif py == ey:
                px += ix
            else:
                px, py = px + ix, py + iy
            e = 0
        elif 0 < d:
            px, e = px + ix, e_x
        else:
            py, e = py + iy, e_y


def astray_scans(n_scans=4, beams=64):
    """Scans (pose with heading 0, unit ranges, the beam's `cos` / `sin` entries carrying the end point's
    offset) whose every beam makes the walk fall back to Bresenham."""
    rs = np.random.RandomState(3)
    tiny = [0, 1e-17, -1e-17, 1e-15, -1e-15, 1e-13, -1e-13, 1e-9, -1e-9, 1e-7, -1e-7, 3e-8, -3e-8, 1e-6, -1e-6]
    out = []
    while len(out) < n_scans:
        x0 = rs.randint(-40, 40) * SCALE * rs.choice([1, 0.5]) + rs.choice(tiny)
        y0 = rs.randint(-40, 40) * SCALE * rs.choice([1, 0.5]) + rs.choice(tiny)
        dxs, dys = [], []
        for _ in range(20000):
            dx = rs.randint(-150, 150) * SCALE + rs.choice(tiny) - x0
            dy = rs.randint(-150, 150) * SCALE + rs.choice(tiny) - y0
            if _walk_goes_astray(SCALE, x0, y0, x0 + 1.0 * dx, y0 + 1.0 * dy):
                dxs.append(dx)
                dys.append(dy)
                if len(dxs) == beams:
                    break
        if len(dxs) == beams:
            out.append((np.array([x0, y0, 0.0]), np.array(dxs), np.array(dys)))
    return out


@pytest.mark.parametrize("path", list(K6_PATHS))
@pytest.mark.parametrize("name", ["mean", "gmapping"])
def test_astray_walks_vs_oracle(pkg, ctx, name, path):
    import pyoracle as po
    set_k6_path(pkg, ctx, path)
    from pyoracle_mapupdate import append_scan_ex
    O = po.Oracle()
    cell_model, rule, st, n_aux = KINDS[name]
    unknown = {0: [0.5], 2: [-1.0, 0.0, 0.0]}[cell_model]
    payload = np.empty((SIZE, SIZE, st))
    payload[:] = unknown
    m = po.GridMapData(cell_model, payload, (SIZE // 2, SIZE // 2), SCALE, unknown)
    aux = np.zeros((SIZE, SIZE, n_aux))
    ctx.upload_map(3, m)
    scans = astray_scans()
    assert len(scans) == 4
    for pose, c, s in scans:
        rng = np.ones(c.size)
        tr = po.ScanData(rng, np.zeros(c.size), None, None, po.TRIG_CACHED, 0.0, 1.0, s, c)
        tr.angle = np.arange(c.size, dtype=np.float64)
        nu_o = append_scan_ex(O, m, aux, rule, pose, rng, tr.angle, None, quality=0.8, blur=0.0, trig=tr)
        nu = ctx.map_append_scan(3, rule, pose, rng, c, s, None, quality=0.8, blur=0.0)
        assert nu == nu_o
        got = ctx.map_download_window(3, 0, 0, SIZE, SIZE, st)
        np.testing.assert_array_equal(got[..., 0], m.payload[..., 0])
        np.testing.assert_allclose(got, m.payload, rtol=1e-13, atol=1e-15)
        np.testing.assert_array_equal(ctx.map_download_aux(3, 0, 0, SIZE, SIZE, n_aux), aux)
    ctx.map_release(3)


def vertex_grazing_scans(n_scans=6, beams=192):
    """Rays aimed at grid vertices with a perpendicular miss between 1e-10 and 1e-5 m, log-uniform, on either side:
    the tie test of world_to_cells (|d| <= 1e-7) lands on both sides of its tolerance, at one vertex or at several (a
    ray through one vertex from a rational offset passes others close by).  Ranges reach 2 ... 40 cells beyond the
    vertex."""
    rs = np.random.RandomState(77)
    out = []
    for k in range(n_scans):
        pose = np.array([rs.uniform(-2, 2), rs.uniform(-2, 2), rs.uniform(-np.pi, np.pi)])
        if k % 2 == 0:  # robot on a cell centre: rational directions meet many vertices
            pose[:2] = (np.floor(pose[:2] / SCALE) + 0.5) * SCALE
        ang, rng = [], []
        for _ in range(beams):
            v = (np.floor(pose[:2] / SCALE) + rs.randint(-60, 61, 2)) * SCALE  # a grid vertex
            d = v - pose[:2]
            dist = np.hypot(*d)
            if dist < 2 * SCALE:
                continue
            miss = rs.choice([-1.0, 1.0]) * 10.0 ** rs.uniform(-10, -5)
            a = np.arctan2(d[1], d[0]) + miss / dist
            ang.append(a - pose[2])
            rng.append(dist + rs.uniform(2, 40) * SCALE)
        out.append((pose, np.array(rng), np.array(ang), (rs.rand(len(rng)) < 0.8).astype(np.int32),
                    [0.0, 0.15][k % 2], np.inf))
    return out


@pytest.mark.parametrize("path", ["gather", "counting"])
@pytest.mark.parametrize("name", ["mean", "gmapping"])
def test_vertex_grazing_rays_vs_oracle(pkg, ctx, name, path):
    """The walks of beams that pass grid vertices within the tie tolerance or just outside it (mu_walk_beam_wave: pieces
    of the closed form between ties, classification clear of the tolerance by 1e-9, everything else step by step):
    cell updates, payload and counters against the oracle's sequential walk."""
    import pyoracle as po
    from pyoracle_mapupdate import append_scan_ex
    set_k6_path(pkg, ctx, path)
    O = po.Oracle()
    cell_model, rule, st, n_aux = KINDS[name]
    unknown = {0: [0.5], 1: [1.0, 0.0, 0.0, 0.0], 2: [-1.0, 0.0, 0.0]}[cell_model]
    payload = np.empty((SIZE, SIZE, st))
    payload[:] = unknown
    m = po.GridMapData(cell_model, payload, (SIZE // 2, SIZE // 2), SCALE, unknown)
    aux = np.zeros((SIZE, SIZE, max(n_aux, 1)))
    ctx.upload_map(3, m)
    for k, (pose, rng, ang, occ, blur, max_range) in enumerate(vertex_grazing_scans()):
        c, s = pkg.beam_trig(ang)
        tr = po.ScanData(rng, ang, None, None, po.TRIG_CACHED, 0.0, 1.0, s, c)
        tr.angle = np.arange(rng.size, dtype=np.float64)
        nu_o = append_scan_ex(O, m, aux if n_aux else None, rule, pose, rng, tr.angle, occ, quality=0.9, blur=blur,
                              max_range=max_range, trig=tr)
        nu = ctx.map_append_scan(3, rule, pose, rng, c, s, occ, quality=0.9, blur=blur, max_range=max_range)
        assert nu == nu_o, "scan %d" % k
        got = ctx.map_download_window(3, 0, 0, SIZE, SIZE, st)
        if name == "gmapping":
            np.testing.assert_array_equal(got[..., 0], m.payload[..., 0], err_msg="scan %d" % k)
            np.testing.assert_allclose(got, m.payload, rtol=1e-11, atol=1e-13, err_msg="scan %d" % k)
        else:
            np.testing.assert_array_equal(got, m.payload, err_msg="scan %d" % k)
        np.testing.assert_array_equal(ctx.map_download_aux(3, 0, 0, SIZE, SIZE, n_aux), aux[..., :n_aux])
    ctx.map_release(3)


@pytest.mark.parametrize("path", list(K6_PATHS))
@pytest.mark.parametrize("name", list(KINDS))
@pytest.mark.parametrize("estimator", [0, 1])
def test_crafted_scans_vs_oracle(pkg, ctx, name, estimator, path):
    """Ties along diagonals, axis-parallel beams, poses on cell corners and centres, bundles of identical beams: in the
    gather form these are the IRREGULAR beams (sequential walk + bitmap) next to regular ones in the same scan."""
    import pyoracle as po
    set_k6_path(pkg, ctx, path)
    from pyoracle_mapupdate import append_scan_ex
    O = po.Oracle()
    cell_model, rule, st, n_aux = KINDS[name]
    unknown = {0: [0.5], 1: [1.0, 0.0, 0.0, 0.0], 2: [-1.0, 0.0, 0.0]}[cell_model]
    payload = np.empty((SIZE, SIZE, st))
    payload[:] = unknown
    m = po.GridMapData(cell_model, payload, (SIZE // 2, SIZE // 2), SCALE, unknown)
    aux = np.zeros((SIZE, SIZE, max(n_aux, 1)))
    ctx.upload_map(3, m)
    for k, (pose, rng, ang, occ, blur, max_range) in enumerate(crafted_scans()):
        for rep in range(2):  # twice: the second pass meets non-trivial cell states (hits, counters)
            c, s = pkg.beam_trig(ang)
            tr = po.ScanData(rng, ang, None, None, po.TRIG_CACHED, 0.0, 1.0, s, c)
            tr.angle = np.arange(rng.size, dtype=np.float64)
            nu_o = append_scan_ex(O, m, aux if n_aux else None, rule, pose, rng, tr.angle, occ, quality=0.8, blur=blur,
                                  max_range=max_range, trig=tr, est_kind=estimator, shift_amount=0.01 * SCALE)
            nu = ctx.map_append_scan(3, rule, pose, rng, c, s, occ, quality=0.8, blur=blur, max_range=max_range,
                                     estimator=estimator, shift_amount=0.01 * SCALE)
            assert nu == nu_o, "scan %d pass %d" % (k, rep)
            got = ctx.map_download_window(3, 0, 0, SIZE, SIZE, st)
            if estimator == 1 or name == "gmapping":
                # obstacle means / area splits depend continuously on the endpoint: same tolerance as the
                # reference-golden tests; occupancy of const-estimator cells is exact
                if estimator == 0:
                    np.testing.assert_array_equal(got[..., 0], m.payload[..., 0], err_msg="scan %d pass %d" % (k, rep))
                np.testing.assert_allclose(got, m.payload, rtol=1e-11, atol=1e-13, err_msg="scan %d pass %d" % (k, rep))
            else:
                np.testing.assert_array_equal(got, m.payload, err_msg="scan %d pass %d" % (k, rep))
            if n_aux:
                np.testing.assert_array_equal(ctx.map_download_aux(3, 0, 0, SIZE, SIZE, n_aux), aux[..., :n_aux])
    ctx.map_release(3)


def test_non_finite_end_points_are_rejected_or_range_gated(pkg, ctx):
    """no-return beams are commonly inf or NaN.  int(floor(inf / NaN)) differs between host and
    device, and the record buffers are sized on the host: such a beam is refused with a message unless the
    range gate (slam/mapping/max_range) drops it, exactly as the reference would assert in world_to_cell
    (regular_squares_grid.h:42).  Afterwards the same context still updates maps."""
    model, rule, stride, _aux = KINDS["mean"]
    ctx.map_bind(3, model, SIZE, SIZE, (SIZE // 2, SIZE // 2), SCALE, [0.5])
    pose = np.array([0.03, 0.02, 0.1])
    ang = np.deg2rad(np.linspace(-100, 100, 64))
    cos_a, sin_a = pkg.beam_trig(ang)
    good = np.full(64, 5.0)
    n_good = ctx.map_append_scan(3, rule, pose, good, cos_a, sin_a)
    assert n_good > 64 * 40
    for bad_value in (np.inf, -np.inf, np.nan, 1e300):
        rng = good.copy()
        rng[17] = bad_value
        with pytest.raises(pkg.SlamHipError):
            ctx.map_append_scan(3, rule, pose, rng, cos_a, sin_a)  # max_range = infinity: not gated
    # with a finite max_range an infinite range is simply dropped (both sides skip the beam)
    rng = good.copy()
    rng[17] = np.inf
    n_gated = ctx.map_append_scan(3, rule, pose, rng, cos_a, sin_a, max_range=20.0)
    assert 0 < n_gated < n_good
    assert ctx.map_append_scan(3, rule, pose, good, cos_a, sin_a) == n_good
    ctx.map_release(3)


@pytest.mark.parametrize("name", ["mean", "tbm", "gmapping"])
def test_window_grows_by_itself_like_an_unbounded_map(pkg, ctx, name):
    """slamhip_map_set_auto_grow: the same crafted scans appended to a 16x16-cell window that has to grow on every
    side (UnboundedPlainGridMap::ensure_inside, plain_grid_map.h:133-173: old cells keep their external place, new
    area holds the prototype) and to a fixed 400x400 window end in the same cells -- payload and update counters --
    and a window without the switch refuses the scan."""
    cell_model, rule, st, n_aux = KINDS[name]
    unknown = {0: [0.5], 1: [1.0, 0.0, 0.0, 0.0], 2: [-1.0, 0.0, 0.0]}[cell_model]
    ctx.map_bind(3, cell_model, SIZE, SIZE, (SIZE // 2, SIZE // 2), SCALE, unknown)
    ctx.map_bind(4, cell_model, 16, 16, (8, 8), SCALE, unknown)
    scans = crafted_scans()
    pose, rng, ang, occ, blur, max_range = scans[0]
    c, s = pkg.beam_trig(ang)
    with pytest.raises(pkg.SlamHipError):
        ctx.map_append_scan(4, rule, pose, rng, c, s, occ, quality=0.8, blur=blur, max_range=max_range)
    ctx.map_release(4)  # (cells inside the window were updated: start over -- a re-bind alone would keep them)
    ctx.map_bind(4, cell_model, 16, 16, (8, 8), SCALE, unknown)
    ctx.map_set_auto_grow(4, True)
    for pose, rng, ang, occ, blur, max_range in scans:
        c, s = pkg.beam_trig(ang)
        a = ctx.map_append_scan(3, rule, pose, rng, c, s, occ, quality=0.8, blur=blur, max_range=max_range)
        b = ctx.map_append_scan(4, rule, pose, rng, c, s, occ, quality=0.8, blur=blur, max_range=max_range)
        assert a == b
    info = ctx.map_info(4)
    assert info["times_grown"] >= 2 and info["width"] > 16 and info["height"] > 16 and info["cell_model"] == cell_model
    ox, oy = info["origin"]
    # the two windows in external cells: equal where they overlap, the prototype where only one of them reaches
    # (the growth margin of the small one may stick out of the fixed one and vice versa)
    W, H = info["width"], info["height"]
    big = ctx.map_download_window(3, 0, 0, SIZE, SIZE, st)
    small = ctx.map_download_window(4, 0, 0, W, H, st)
    x0, y0 = SIZE // 2 - ox, SIZE // 2 - oy  # fixed-window coordinates of the grown window's cell (0, 0)
    bx0, by0, bx1, by1 = max(x0, 0), max(y0, 0), min(x0 + W, SIZE), min(y0 + H, SIZE)
    assert bx1 - bx0 > 150 and by1 - by0 > 150
    np.testing.assert_array_equal(small[by0 - y0:by1 - y0, bx0 - x0:bx1 - x0], big[by0:by1, bx0:bx1])
    outside = np.ones((SIZE, SIZE), bool)
    outside[by0:by1, bx0:bx1] = False
    assert (big[outside] == np.array(unknown)).all()
    inside = np.zeros((H, W), bool)
    inside[by0 - y0:by1 - y0, bx0 - x0:bx1 - x0] = True
    assert (small[~inside] == np.array(unknown)).all()
    if n_aux:
        np.testing.assert_array_equal(ctx.map_download_aux(4, bx0 - x0, by0 - y0, bx1 - bx0, by1 - by0, n_aux),
                                      ctx.map_download_aux(3, bx0, by0, bx1 - bx0, by1 - by0, n_aux))
    ctx.map_release(3)
    ctx.map_release(4)


@pytest.mark.parametrize("name", ["mean", "tbm", "gmapping"])
def test_queued_updates_equal_awaited_updates(pkg, ctx, name):
    """slamhip_map_set_deferred / slamhip_map_drain: the crafted scans, six rounds of them (so the ring of 64 queued
    updates wraps), appended without waiting -- every call returns -1 -- into a window that grows by itself, against
    the same scans awaited one by one: the drained update count is the sum, the windows are equal byte for byte, a
    download in between is ordered behind what is queued, and a beam outside a window that may not grow surfaces at
    the drain."""
    cell_model, rule, st, n_aux = KINDS[name]
    unknown = {0: [0.5], 1: [1.0, 0.0, 0.0, 0.0], 2: [-1.0, 0.0, 0.0]}[cell_model]
    for mid in (3, 4):
        ctx.map_bind(mid, cell_model, 48, 48, (24, 24), SCALE, unknown)
        ctx.map_set_auto_grow(mid, True)
    scans = crafted_scans() * 6
    total = 0
    for pose, rng, ang, occ, blur, max_range in scans:
        c, s = pkg.beam_trig(ang)
        total += ctx.map_append_scan(3, rule, pose, rng, c, s, occ, quality=0.8, blur=blur, max_range=max_range)
    ctx.map_set_deferred(True)
    mid_way = None
    for k, (pose, rng, ang, occ, blur, max_range) in enumerate(scans):
        c, s = pkg.beam_trig(ang)
        assert ctx.map_append_scan(4, rule, pose, rng, c, s, occ, quality=0.8, blur=blur, max_range=max_range) == -1
        if k == len(scans) // 2:
            i4 = ctx.map_info(4)
            mid_way = ctx.map_download_window(4, 0, 0, i4["width"], i4["height"], st)  # behind the queued updates
    drained = ctx.map_drain()
    assert ctx.map_drain() == 0
    ctx.map_set_deferred(False)
    assert len(scans) > 64 and mid_way is not None and np.isfinite(mid_way).all()
    assert drained == total  # (the ring collected itself once on the way: its count is carried along)
    i3, i4 = ctx.map_info(3), ctx.map_info(4)
    assert (i3["width"], i3["height"], i3["origin"]) == (i4["width"], i4["height"], i4["origin"])
    a = ctx.map_download_window(3, 0, 0, i3["width"], i3["height"], st)
    b = ctx.map_download_window(4, 0, 0, i4["width"], i4["height"], st)
    assert a.tobytes() == b.tobytes()
    if n_aux:
        assert ctx.map_download_aux(3, 0, 0, i3["width"], i3["height"], n_aux).tobytes() == \
            ctx.map_download_aux(4, 0, 0, i4["width"], i4["height"], n_aux).tobytes()
    # a window that may not grow: the failure of a queued update is reported by the drain
    ctx.map_release(4)
    ctx.map_bind(4, cell_model, 16, 16, (8, 8), SCALE, unknown)
    ctx.map_set_deferred(True)
    pose, rng, ang, occ, blur, max_range = scans[0]
    c, s = pkg.beam_trig(ang)
    assert ctx.map_append_scan(4, rule, pose, rng, c, s, occ, quality=0.8, blur=blur, max_range=max_range) == -1
    with pytest.raises(pkg.SlamHipError):
        ctx.map_drain()
    ctx.map_set_deferred(False)
    ctx.map_release(3)
    ctx.map_release(4)


@pytest.mark.parametrize("path", list(K6_PATHS))
def test_long_tbm_chains_through_every_regime_vs_oracle(pkg, ctx, path):
    """r06: a TbmBaseCell chain of a wave's round (up to 64 observations of one cell, mu_wave_apply) runs WITHOUT per-update
    tests once the round has established a zero conflict mass, observation unknown >= 0.5 and masses away from the
    subnormal range -- the conflict mass as one addition, the seven quotients of normalize / normalize_conflict from two
    reciprocals refined ahead (no scaling, no fix-up: tools/probes/tbm_div_probe.hip).  Here the robot's own cell and its
    neighbours take ~1000 updates per scan, scan after scan from ONE pose, through every regime the tests separate:
      * small qualities (the fast round), the unknown mass decaying geometrically -- 0.955^n: below 2^-300 after ~4500
        updates, where the round hands over to the plain form, and on through the subnormals to an exact zero;
      * qualities above 0.5 (observation unknown < 0.5: the plain form from the start);
      * occupied and empty observations mixed on the same cells (a conflict share in every update), a beam quality per
        point, an uploaded cell that carries a conflict mass and one with a negative zero.
    After every scan: all four belief masses of every cell, bit for bit with the oracle's (map_update_oracle.c)."""
    import pyoracle as po
    from pyoracle_mapupdate import RULE_TBM, append_scan_q
    set_k6_path(pkg, ctx, path)
    O = po.Oracle()
    size = 120
    unknown = [1.0, 0.0, 0.0, 0.0]
    payload = np.empty((size, size, 4))
    payload[:] = unknown
    rs = np.random.RandomState(17)
    # a few cells near the robot start from something else: a conflict mass, a negative zero, a tiny mass
    payload[60, 61] = [0.2, 0.3, 0.4, 0.1]
    payload[61, 60] = [1.0, -0.0, 0.0, 0.0]
    payload[59, 60] = [1e-200, 0.5, 0.5 - 1e-200, 0.0]
    m = po.GridMapData(1, payload, (size // 2, size // 2), SCALE, unknown)
    ctx.upload_map(3, m)
    n = 1000
    ang = np.linspace(-np.pi, np.pi, n, endpoint=False)
    pose = np.array([SCALE / 2, SCALE / 2, 0.3])
    c, s = pkg.beam_trig(ang)
    regimes = [("small qualities", (0.95, 0.04, 0.01, 0.045), 1.0, 12), ("high quality", (0.95, 0.9, 0.01, 0.8), 1.0, 2),
               ("mid", (0.9, 0.45, 0.05, 0.5), 1.0, 3), ("small again", (0.95, 0.04, 0.01, 0.045), 1.0, 6)]
    k = 0
    for name, base, quality, scans in regimes:
        for _ in range(scans):
            rng = 0.3 + 2.5 * rs.rand(n)  # end points at every distance: hits and passes mix on the cells around the robot
            occ = (rs.rand(n) < 0.8).astype(np.int32)
            bq = rs.uniform(0.5, 1.0, n) if k % 3 == 2 else None
            tr = po.ScanData(rng, np.arange(n, dtype=np.float64), None, None, po.TRIG_CACHED, 0.0, 1.0, s, c)
            nu_o = append_scan_q(O, m, None, RULE_TBM, pose, rng, tr.angle, bq if bq is not None else np.ones(n), occ,
                                 quality=quality, base=base, blur=0.05, trig=tr)
            nu = ctx.map_append_scan(3, pkg.RULE_TBM, pose, rng, c, s, occ, quality=quality, base=base, blur=0.05,
                                     beam_quality=bq)
            assert nu == nu_o
            got = ctx.map_download_window(3, 0, 0, size, size, 4)
            bad = np.nonzero(got.view(np.uint64) != m.payload.view(np.uint64))
            assert len(bad[0]) == 0, (name, k, bad[0][:4], bad[1][:4], got[bad][:4], m.payload[bad][:4])
            k += 1
    robot = m.payload[60, 60]
    assert robot[3] == 0.0 and robot[0] < 1e-60 and abs(robot.sum() - 1.0) < 1e-12  # (the unknown mass did decay)
    ctx.map_release(3)
