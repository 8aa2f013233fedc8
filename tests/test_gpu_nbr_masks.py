"""GPU suite: the neighbourhood masks of a dense GMAPPING window (csrc/slamhip_internal.h MapView, gm_score_device.h).

The GMapping scorer reads, per beam, ONE mask that says which cells of its 3 x 3 window are full, kept in the pad of
the centre cell; every writer of the window has to keep those masks true.  Checked here after every kind of write:
the masks equal the ones the occupancies give (the testing library's slamhip_map_debug_nbr_masks) and the scores equal
the ones of a second context that got the same cells as a fresh upload (whose masks are derived from scratch) -- bit for
bit, and equal to the oracle's within the scorer's bar.  The scorer's arithmetic itself is pinned where it always was
(test_gpu_parity.py: goldens of the compiled reference)."""
import ctypes as C

import numpy as np
import pytest

import __graft_entry__ as ge

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    return ge.load_package()


@pytest.fixture(scope="module")
def tctx(pkg):
    c = pkg.Context(0, testing=True)
    yield c
    c.close()


@pytest.fixture(scope="module")
def fresh(pkg):
    c = pkg.Context(0, testing=True)
    yield c
    c.close()


def masks(ctx, map_id):
    valid, bad = C.c_int(-1), C.c_longlong(-1)
    rc = ctx.L.slamhip_map_debug_nbr_masks(ctx.h, map_id, C.byref(valid), C.byref(bad))
    assert rc == 0
    return valid.value, bad.value


@pytest.fixture(scope="module")
def po():
    import pyoracle
    return pyoracle


def score(ctx, map_id, cfg, poses):
    ctx.gm_cache_reset()  # (the OOPE's cross-pose cache lives in the context: every sequence starts from an empty one)
    return ctx.score_poses(map_id, cfg, poses)


def poses_around(p, n=48, seed=3):
    r = np.random.default_rng(seed)
    out = np.tile(np.asarray(p, dtype=np.float64), (n, 1))
    out[:, :2] += r.uniform(-0.3, 0.3, (n, 2))
    out[:, 2] += r.uniform(-0.1, 0.1, n)
    return out


def scores_of_fresh_upload(pkg, fresh, tctx, map_id, scan, cfg, poses, unknown):
    """the window of `tctx`'s map downloaded and uploaded to another context as new: its masks come from one pass"""
    info = tctx.map_info(map_id)
    w, h = info["width"], info["height"]
    cells = tctx.map_download_window(map_id, 0, 0, w, h, 3)
    fresh.map_bind(1, pkg.CELL_GMAPPING, w, h, info["origin"], info["scale"], unknown)
    fresh.map_upload_window(1, 0, 0, cells)
    c, s = pkg.beam_trig(scan.angle)
    fresh.scan_upload(scan.range, c, s, scan.weight)
    out = score(fresh, 1, cfg, poses)
    fresh.map_release(1)
    return out


def test_masks_follow_every_writer(pkg, tctx, fresh):
    from synth import make_scene
    sc = make_scene(cell_model=2, size=600, scale=0.05, n_beams=720, seed=5)
    m, scan = sc["map"], sc["scan"]
    cfg = pkg.spe_cfg(oope=pkg.OOPE_GMAPPING)
    c, s = pkg.beam_trig(scan.angle)
    poses = poses_around(sc["true_pose"])
    tctx.upload_map(0, m)
    assert masks(tctx, 0) == (0, 0)  # nobody asked yet
    tctx.scan_upload(scan.range, c, s, scan.weight)
    base = score(tctx, 0, cfg, poses)
    assert masks(tctx, 0) == (1, 0)
    np.testing.assert_array_equal(base, scores_of_fresh_upload(pkg, fresh, tctx, 0, scan, cfg, poses, m.unknown))

    # K6 (GMapping rule) from poses that draw new walls and wear old ones down: cells change sides both ways
    r = np.random.default_rng(9)
    for k, path in enumerate([0, 1, 2, 0, 1, 2]):
        tctx.set_option(pkg.OPT_K6_PATH, path)
        pose = sc["true_pose"] + np.array([r.uniform(-1.0, 1.0), r.uniform(-1.0, 1.0), r.uniform(-0.5, 0.5)])
        rng = np.minimum(scan.range, 14.0) * r.uniform(0.4, 0.85)  # hits in what the map holds as free space
        nu = tctx.map_append_scan(0, pkg.RULE_GMAPPING, pose, rng, c, s, None)
        assert nu > 1000
        assert masks(tctx, 0) == (1, 0), "after update %d (path %d)" % (k, path)
        tctx.scan_upload(scan.range, c, s, scan.weight)
        got = score(tctx, 0, cfg, poses)
        np.testing.assert_array_equal(got, scores_of_fresh_upload(pkg, fresh, tctx, 0, scan, cfg, poses, m.unknown))
        assert not np.array_equal(got, base)
    tctx.set_option(pkg.OPT_K6_PATH, 0)

    # the host's dirty log: full cells made free, free cells made full, on the rim too
    cells = tctx.map_download_window(0, 0, 0, m.width, m.height, 3)
    full = np.argwhere(cells[..., 0] >= 0.1)
    free = np.argwhere(cells[..., 0] < 0.1)
    pick_full = full[r.choice(len(full), min(300, len(full)), replace=False)]
    pick_free = free[r.choice(len(free), 300, replace=False)]
    xy = np.concatenate([pick_full[:, ::-1], pick_free[:, ::-1], [[0, 0], [m.width - 1, m.height - 1], [0, 7]]])
    vals = np.zeros((len(xy), 3))
    vals[:len(pick_full)] = [0.0, 0.0, 0.0]
    vals[len(pick_full):] = [0.9, 0.0, 0.0]
    vals[len(pick_full):, 1:] = r.uniform(-10, 10, (len(xy) - len(pick_full), 2))
    tctx.map_apply_dirty(0, xy, vals)
    assert masks(tctx, 0) == (1, 0)
    got = score(tctx, 0, cfg, poses)
    np.testing.assert_array_equal(got, scores_of_fresh_upload(pkg, fresh, tctx, 0, scan, cfg, poses, m.unknown))

    # a window uploaded over a part of the map (the rim of the upload borders cells that stay)
    patch = np.zeros((40, 50, 3))
    patch[..., 0] = r.choice([0.0, 0.05, 0.1, 0.8], (40, 50))
    patch[..., 1:] = r.uniform(-5, 5, (40, 50, 2))
    tctx.map_upload_window(0, 280, 290, patch)
    tctx.map_upload_window(0, 0, 0, patch)
    tctx.map_upload_window(0, m.width - 50, m.height - 40, patch)
    assert masks(tctx, 0) == (1, 0)
    got = score(tctx, 0, cfg, poses)
    np.testing.assert_array_equal(got, scores_of_fresh_upload(pkg, fresh, tctx, 0, scan, cfg, poses, m.unknown))

    # another threshold: its FIRST call is served by the nine-cell form (the window keeps the masks it has: two scorer
    # configurations alternating on one map must not re-derive 0.5 GB of masks per call); a threshold that
    # stays -- asks again -- takes the masks over.  The same scores all three ways.
    cfg2 = pkg.spe_cfg(oope=pkg.OOPE_GMAPPING, gm_th=0.5)
    got = score(tctx, 0, cfg2, poses)
    assert masks(tctx, 0) == (1, 0)
    want2 = scores_of_fresh_upload(pkg, fresh, tctx, 0, scan, cfg2, poses, m.unknown)
    np.testing.assert_array_equal(got, want2)
    np.testing.assert_array_equal(score(tctx, 0, cfg, poses), scores_of_fresh_upload(pkg, fresh, tctx, 0, scan, cfg, poses, m.unknown))
    for _ in range(2):
        np.testing.assert_array_equal(score(tctx, 0, cfg2, poses), want2)
        assert masks(tctx, 0) == (1, 0)
    tctx.map_release(0)


def test_masks_of_a_window_that_grows(pkg, tctx, fresh):
    """An update beyond the window re-binds it (slamhip_map_set_auto_grow): the masks are dropped with the old window
    and derived again by the next scorer call."""
    from synth import make_scene
    sc = make_scene(cell_model=2, size=300, scale=0.05, n_beams=360, seed=6)
    m, scan = sc["map"], sc["scan"]
    cfg = pkg.spe_cfg(oope=pkg.OOPE_GMAPPING)
    c, s = pkg.beam_trig(scan.angle)
    poses = poses_around(sc["true_pose"], n=16)
    tctx.upload_map(0, m)
    tctx.map_set_auto_grow(0, True)
    tctx.scan_upload(scan.range, c, s, scan.weight)
    score(tctx, 0, cfg, poses)
    assert masks(tctx, 0) == (1, 0)
    far = np.full(scan.range.size, 12.0)  # 240 cells: beyond the 300-cell window's rim
    tctx.map_append_scan(0, pkg.RULE_GMAPPING, sc["true_pose"], far, c, s, None)
    assert tctx.map_info(0)["times_grown"] >= 1
    assert masks(tctx, 0)[0] == 0
    tctx.scan_upload(scan.range, c, s, scan.weight)
    got = score(tctx, 0, cfg, poses)
    assert masks(tctx, 0) == (1, 0)
    np.testing.assert_array_equal(got, scores_of_fresh_upload(pkg, fresh, tctx, 0, scan, cfg, poses, m.unknown))
    # ... and kept from there on
    tctx.map_append_scan(0, pkg.RULE_GMAPPING, sc["true_pose"] + [0.5, 0.2, 0.1], scan.range * 0.7, c, s, None)
    assert masks(tctx, 0) == (1, 0)
    tctx.map_release(0)


def test_scores_with_masks_equal_the_oracles(pkg, tctx, oracle, po):
    """the scorer through the masks against the CPU restatement, on a map K6 has written to"""
    from synth import make_scene
    sc = make_scene(cell_model=2, size=600, scale=0.05, n_beams=720, seed=7)
    m, scan = sc["map"], sc["scan"]
    cfg = pkg.spe_cfg(oope=pkg.OOPE_GMAPPING, pose_trig=1)
    c, s = pkg.beam_trig(scan.angle)
    poses = poses_around(sc["true_pose"], n=32)
    tctx.upload_map(0, m)
    tctx.scan_upload(scan.range, c, s, scan.weight)
    score(tctx, 0, cfg, poses)
    tctx.map_append_scan(0, pkg.RULE_GMAPPING, sc["true_pose"] + [0.8, -0.6, 0.2], np.minimum(scan.range, 14.0) * 0.6, c, s,
                         None)
    assert masks(tctx, 0) == (1, 0)
    tctx.scan_upload(scan.range, c, s, scan.weight)
    got = score(tctx, 0, cfg, poses)
    m.payload[...] = tctx.map_download_window(0, 0, 0, m.width, m.height, 3)
    want = oracle.score_poses(m, scan, po.make_cfg(oope=po.OOPE_GMAPPING), poses, po.Oracle.new_gm_cache())
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=0)
    tctx.map_release(0)


@pytest.mark.parametrize("mode", ["fast", "sorted"])
def test_tile_pool_masks_and_settle_states_follow_the_filter(pkg, oracle, mode):
    """Per-particle copy-on-write maps (tile_pool.h): a tile's masks know the cells of their own tile; the batched map
    update, copy-on-write clones and resampling keep them -- and the pool's settle states (two bits per cell that the
    update's free-space fast path reads instead of the cell).  The filter of test_gpu_particle_maps.py (every step
    against the oracle: poses, weights, every particle's map) with masks and states checked after every step."""
    from test_gpu_particle_maps import run_both
    options = {"fast": (), "sorted": ((pkg.OPT_K6_BATCH_FAST, 0),)}[mode]
    # (the run of test_particle_maps_resampling_shares_then_clones_tiles: it resamples -- duplicates share tiles, the next
    # update clones what it writes)
    pf, log, _ = run_both(pkg, oracle, n=8, seed0=3000, n_steps_extra=20, options=options, check_masks=True,
                          gp=[0, 0.1, 0, 0.05, 0, 0, 0, 0])
    assert sum(1 for res, _ in log if res) >= 1 and log[-1][1]["cow_copies"] > 0


@pytest.mark.parametrize("th", [0.1, 0.0, -2.0, 0.5])
def test_masks_at_odd_thresholds_and_cells(pkg, tctx, oracle, po, th):
    """What `full` means is the scorer's own test, !(prob_occ < th): a NaN occupancy is full, at th <= 0 free and
    never-observed cells are full too (every mask is all ones), the rim of the window and the cells beyond it take the
    nine-cell form.  Scores through the masks against the CPU restatement, poses near the window's rim included."""
    from synth import make_scene
    sc = make_scene(cell_model=2, size=400, scale=0.05, n_beams=540, seed=8)
    m, scan = sc["map"], sc["scan"]
    r = np.random.default_rng(21)
    holes = r.integers(0, m.width, (400, 2))
    m.payload[holes[:, 1], holes[:, 0], 0] = np.nan  # NaN occupancies (their obstacle means stay finite)
    m.payload[:3, :, 0] = 0.9   # walls on the window's rim
    m.payload[:, -2:, 0] = 0.9
    cfg = pkg.spe_cfg(oope=pkg.OOPE_GMAPPING, gm_th=th, pose_trig=1)
    c, s = pkg.beam_trig(scan.angle)
    poses = np.concatenate([poses_around(sc["true_pose"], n=24, seed=4),
                            poses_around(sc["true_pose"] + [6.0, -7.5, 0.4], n=8, seed=5)])  # beams that leave the window
    tctx.upload_map(0, m)
    tctx.scan_upload(scan.range, c, s, scan.weight)
    got = score(tctx, 0, cfg, poses)
    assert masks(tctx, 0) == (1, 0)
    want = oracle.score_poses(m, scan, po.make_cfg(oope=po.OOPE_GMAPPING, gm_th=th), poses, po.Oracle.new_gm_cache())
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=0)
    tctx.map_release(0)


@pytest.mark.parametrize("window", [0, 2])
def test_other_window_sizes_take_the_cell_loop(pkg, tctx, oracle, po, window):
    """slam/scmtch/oope/window other than 1 (no shipped configuration has one): the generic loop over (2 w + 1)^2 cells, no
    masks -- against the CPU restatement, on a map whose masks exist (a window-1 scorer ran before)."""
    from synth import make_scene
    sc = make_scene(cell_model=2, size=400, scale=0.05, n_beams=540, seed=9)
    m, scan = sc["map"], sc["scan"]
    c, s = pkg.beam_trig(scan.angle)
    poses = poses_around(sc["true_pose"], n=24, seed=6)
    tctx.upload_map(0, m)
    tctx.scan_upload(scan.range, c, s, scan.weight)
    score(tctx, 0, pkg.spe_cfg(oope=pkg.OOPE_GMAPPING, pose_trig=1), poses)  # (masks derived)
    got = score(tctx, 0, pkg.spe_cfg(oope=pkg.OOPE_GMAPPING, gm_window=window, pose_trig=1), poses)
    want = oracle.score_poses(m, scan, po.make_cfg(oope=po.OOPE_GMAPPING, gm_window=window), poses, po.Oracle.new_gm_cache())
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=0)
    tctx.map_release(0)
