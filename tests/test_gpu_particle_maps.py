"""GPU suite, SURVEY 8f N2: GMapping filter with per-particle copy-on-write maps (device tile pool).

There is no reference run for this mode -- the reference revision shares one map object among all
particles (Q20) -- so parity is against the oracle running the same filter with one private dense map
per particle (oracle/slam_oracle.c: orc_gmapping_set_particle_maps; the oracle's shared-map mode is
pinned to the compiled reference by tests/test_oracle_mapupdate.py).  The sharing semantics themselves
(copy = table copy, write clones a shared tile, untouched area is one unknown tile) are the
reference's LazyTiledGridMap rules and are checked as invariants of the pool statistics.

The common ancestor map is the first scan appended from the true pose (an empty one gives every
particle probability 0 and NaN weights -- the reference's shared map only escapes that because
particle 0 writes before particle 1 reads)."""
import os

import numpy as np
import pytest
from helpers import load

import __graft_entry__ as ge

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    return ge.load_package()


def run_both(pkg, oracle, n_steps_extra=0, n=8, seed0=2000, gp=None, adder=None, options=(), check_masks=False):
    import pyoracle as po
    from pyoracle_mapupdate import (RULE_GMAPPING, append_scan_ex, gmapping_enable_particle_maps,
                                    gmapping_particle_map)
    g = load("gmapping_pf_update.npz")
    w, h = [int(v) for v in g["size"]]
    scale = float(g["scale"])
    unknown = g["unknown"][:3]
    gp = g["gp"] if gp is None else np.asarray(gp, dtype=np.float64)
    seeds = np.arange(seed0, seed0 + n, dtype=np.uint32)
    ox, oy = [int(v) for v in g["origin"]]
    r0, a0, pose0 = g["step0_range"], g["step0_angle"], g["step0_delta"]
    # HIP side: dense ancestor (K6, pinned to the reference elsewhere) -> tile pool
    ctx = pkg.Context(0, testing=check_masks)  # (check_masks: tests/test_gpu_nbr_masks.py, a hook of the testing library)
    for opt, val in options:
        ctx.set_option(opt, val)
    ctx.map_bind(4, 2, w, h, g["origin"], scale, unknown)
    c0, s0 = pkg.beam_trig(a0)
    ctx.map_append_scan(4, pkg.RULE_GMAPPING, pose0, r0, c0, s0)
    pf = pkg.GmappingFilter(ctx, pkg.gmapping_params(gp8=gp, skip_rate=3, pose_trig=1), n, seeds)
    adder = adder or {}
    pf.enable_particle_maps(4, extent_tiles=8, pool_tiles=16 + 24 * n, **adder)
    # oracle side
    payload = np.tile(unknown, (h, w, 1)).astype(np.float64)
    m = po.GridMapData(po.CELL_GMAPPING, payload, g["origin"], scale, unknown)
    aux = np.zeros((h, w, 2))
    append_scan_ex(oracle, m, aux, RULE_GMAPPING, pose0, r0, a0)
    opf = oracle.gmapping_create(n, gp, seeds, skip_rate=3)
    gmapping_enable_particle_maps(oracle, opf, m, aux, **{{"estimator": "est_kind"}.get(k, k): v for k, v in adder.items()})
    got_p, got_a = pf.particle_map(n // 2, -ox, -oy, w, h)  # the ancestor went into the tiles intact
    np.testing.assert_array_equal(got_p[..., 0], m.payload[..., 0])
    np.testing.assert_array_equal(got_a, aux)

    n_base = int(g["n_steps"])
    steps = list(range(n_base)) + [1 + (k % (n_base - 1)) for k in range(n_steps_extra)]  # replay scans
    log = []
    for it, k in enumerate(steps):
        extra = np.arange(9000 + 100 * it, 9000 + 100 * it + n, dtype=np.uint32)
        rng, ang, d = g["step%d_range" % k], g["step%d_angle" % k], g["step%d_delta" % k]
        res, idx = pf.step(4, rng, ang, None, d, 7 + it)
        ores, oidx = opf.step(m, rng, ang, None, d, 7 + it, extra)
        poses, wts, ms = pf.state()
        oposes, owts, oms = opf.state()
        assert res == ores, it
        if res:
            np.testing.assert_array_equal(idx, oidx)
        assert np.all(np.isfinite(owts))
        np.testing.assert_array_equal(ms, oms)
        np.testing.assert_allclose(poses, oposes, rtol=0, atol=1e-10, err_msg="step %d" % it)
        np.testing.assert_allclose(wts, owts, rtol=1e-9, atol=0)
        for i in range(n):
            got_p, got_a = pf.particle_map(i, -ox, -oy, w, h)
            want_p, want_a = gmapping_particle_map(oracle, opf, i)
            np.testing.assert_array_equal(got_p[..., 0], want_p[..., 0], err_msg="step %d particle %d" % (it, i))
            np.testing.assert_allclose(got_p[..., 1:], want_p[..., 1:], rtol=1e-12, atol=1e-14)
            np.testing.assert_array_equal(got_a, want_a)
        if check_masks:
            import ctypes as C
            valid, bad = C.c_int(-1), C.c_longlong(-1)
            assert ctx.L.slamhip_gmapping_debug_nbr_masks(pf.h, C.byref(valid), C.byref(bad)) == 0
            assert (valid.value, bad.value) == (1, 0), "step %d" % it
            assert ctx.L.slamhip_gmapping_debug_settle_states(pf.h, C.byref(bad)) == 0
            assert bad.value == 0, "settle states, step %d" % it
        log.append((res, pf.particle_map_stats()))
    return pf, log, (ox, oy, w, h)


@pytest.mark.parametrize("mode", ["fast", "sorted", "key64"])
def test_particle_maps_filter_vs_oracle(pkg, oracle, mode):
    """The batched map update three ways: `fast` (default) settles the free observations of zero-mean cells with
    atomics and sorts only the rest (k_mu_classify); `sorted` (SLAMHIP_OPT_K6_BATCH_FAST = 0) sorts every record into cell
    chains; `key64` is the sorted pipeline with 8-byte (particle, cell) keys -- the fall-back for batches whose
    particle and key-window bits exceed 32 -- forced through SLAMHIP_OPT_K6_BATCH_KEY64."""
    options = {"fast": (), "key64": ((pkg.OPT_K6_BATCH_KEY64, 1),), "sorted": ((pkg.OPT_K6_BATCH_FAST, 0),)}[mode]
    n = 8
    pf, log, (ox, oy, w, h) = run_both(pkg, oracle, n=n, options=options)
    st = log[-1][1]
    assert st["cell_updates"] > 0 and st["tiles_in_use"] > 16
    # the maps of different particles really differ (pose noise -> different cells updated)
    a, _ = pf.particle_map(0, -ox, -oy, w, h)
    b, _ = pf.particle_map(n - 1, -ox, -oy, w, h)
    assert np.count_nonzero(a[..., 0] != b[..., 0]) > 0


def test_particle_maps_grow_with_the_scans(pkg, oracle):
    """UnboundedLazyTiledGridMap grows when a scan reaches beyond it (lazy_tiled_grid_map.h:128-187).  Here
    the ancestor is a small dense window holding a range-gated first scan (2.5 m), the tile extent starts as
    ONE tile (6.4 m across), and the filter's own updates (no gate) reach further: the extent must grow mid-run, in
    front of every later lookup (K3 through the tile tables, K6, download), without changing a result."""
    import pyoracle as po
    from pyoracle_mapupdate import RULE_GMAPPING, append_scan_ex, gmapping_enable_particle_maps, gmapping_particle_map
    g = load("gmapping_pf_update.npz")
    w, h = [int(v) for v in g["size"]]
    scale = float(g["scale"])
    unknown = g["unknown"][:3]
    gp, n = g["gp"], 6
    seeds = np.arange(3000, 3000 + n, dtype=np.uint32)
    ox, oy = [int(v) for v in g["origin"]]
    r0, a0, pose0 = g["step0_range"], g["step0_angle"], g["step0_delta"]
    small, gate = 128, 2.5
    ctx = pkg.Context(0)
    ctx.map_bind(4, 2, small, small, [small // 2, small // 2], scale, unknown)
    c0, s0 = pkg.beam_trig(a0)
    ctx.map_append_scan(4, pkg.RULE_GMAPPING, pose0, r0, c0, s0, max_range=gate)
    pf = pkg.GmappingFilter(ctx, pkg.gmapping_params(gp8=gp, skip_rate=3, pose_trig=1), n, seeds)
    pf.enable_particle_maps(4, extent_tiles=1, pool_tiles=16 + 40 * n)
    payload = np.tile(unknown, (h, w, 1)).astype(np.float64)
    m = po.GridMapData(po.CELL_GMAPPING, payload, g["origin"], scale, unknown)
    aux = np.zeros((h, w, 2))
    append_scan_ex(oracle, m, aux, RULE_GMAPPING, pose0, r0, a0, max_range=gate)
    opf = oracle.gmapping_create(n, gp, seeds, skip_rate=3)
    gmapping_enable_particle_maps(oracle, opf, m, aux)
    tiles0 = pf.particle_map_stats()["tiles_in_use"]
    for it in range(int(g["n_steps"])):
        extra = np.arange(9000 + 100 * it, 9000 + 100 * it + n, dtype=np.uint32)
        rng, ang, d = g["step%d_range" % it], g["step%d_angle" % it], g["step%d_delta" % it]
        res, idx = pf.step(4, rng, ang, None, d, 7 + it)
        ores, oidx = opf.step(m, rng, ang, None, d, 7 + it, extra)
        assert res == ores
        if res:
            np.testing.assert_array_equal(idx, oidx)
        poses, wts, ms = pf.state()
        oposes, owts, oms = opf.state()
        np.testing.assert_allclose(poses, oposes, rtol=0, atol=1e-10, err_msg="step %d" % it)
        np.testing.assert_allclose(wts, owts, rtol=1e-9, atol=0)
        for i in range(n):
            got_p, got_a = pf.particle_map(i, -ox, -oy, w, h)  # the oracle's window: beyond the first extent
            want_p, want_a = gmapping_particle_map(oracle, opf, i)
            np.testing.assert_array_equal(got_p[..., 0], want_p[..., 0], err_msg="step %d particle %d" % (it, i))
            np.testing.assert_allclose(got_p[..., 1:], want_p[..., 1:], rtol=1e-12, atol=1e-14)
            np.testing.assert_array_equal(got_a, want_a)
    # cells were written beyond the 128-cell start extent
    far = np.ones((h, w), bool)
    far[oy - 64:oy + 64, ox - 64:ox + 64] = False
    _, a_last = pf.particle_map(0, -ox, -oy, w, h)
    assert np.count_nonzero(a_last[..., 1][far]) > 500
    assert pf.particle_map_stats()["tiles_in_use"] > tiles0


def test_particle_maps_resampling_shares_then_clones_tiles(pkg, oracle):
    """A run long enough to resample twice (the oracle alone was used to find it): right after a
    resampling the duplicates share tiles and nothing was copied for it; the next update clones
    only what it writes."""
    pf, log, _ = run_both(pkg, oracle, n_steps_extra=20, n=8, seed0=3000, gp=[0, 0.1, 0, 0.05, 0, 0, 0, 0])
    resampled_at = [i for i, (res, _) in enumerate(log) if res]
    assert len(resampled_at) >= 1, "the sequence was meant to trigger a resampling"
    i = resampled_at[0]
    before, after = log[i - 1][1], log[i][1]
    assert after["tiles_shared"] > 0
    nxt = log[i + 1][1]
    assert nxt["cow_copies"] > after["cow_copies"]  # duplicates wrote -> their tiles were cloned
    assert nxt["tiles_shared"] <= after["tiles_shared"]
    # the pool does not leak: tiles of dropped particles return to the free list
    assert log[-1][1]["tiles_in_use"] <= 16 + 24 * 8


def test_sharded_particle_maps_migrate_on_resampling(pkg):
    """Two shards of the filter (two tile pools, as on two GPUs) against the unsharded filter: the raw
    weights are concatenated (the all-gather), every shard plans the same resampling, maps of
    particles drawn from the other shard travel as exported buffers.  Poses, weights and every
    particle's map must equal the unsharded run bit for bit, through two resamplings."""
    g = load("gmapping_pf_update.npz")
    w, h = [int(v) for v in g["size"]]
    scale, unknown = float(g["scale"]), g["unknown"][:3]
    ox, oy = [int(v) for v in g["origin"]]
    n, half = 8, 4
    gp = [0, 0.1, 0, 0.05, 0, 0, 0, 0]
    seeds = np.arange(3000, 3000 + n, dtype=np.uint32)
    ctx = pkg.Context(0)
    ctx.map_bind(4, 2, w, h, g["origin"], scale, unknown)
    c0, s0 = pkg.beam_trig(g["step0_angle"])
    ctx.map_append_scan(4, pkg.RULE_GMAPPING, g["step0_delta"], g["step0_range"], c0, s0)
    prm = pkg.gmapping_params(gp8=gp, skip_rate=3, pose_trig=1)
    whole = pkg.GmappingFilter(ctx, prm, n, seeds)
    whole.enable_particle_maps(4, 8, 16 + 24 * n)
    shards = [pkg.GmappingFilter(ctx, prm, n, seeds[r * half:(r + 1) * half], first=r * half, count=half)
              for r in range(2)]
    for s in shards:
        s.enable_particle_maps(4, 8, 16 + 24 * n)
    n_base = int(g["n_steps"])
    steps = list(range(n_base)) + [1 + (k % (n_base - 1)) for k in range(20)]
    resamplings = migrations = 0
    for it, k in enumerate(steps):
        rng, ang, d = g["step%d_range" % k], g["step%d_angle" % k], g["step%d_delta" % k]
        res, idx = whole.step(4, rng, ang, None, d, 7 + it)
        raw = np.concatenate([s.predict_match(4, rng, ang, None, d) for s in shards])
        plans = [s.plan_resample(raw, 7 + it) for s in shards]
        assert plans[0][0] == plans[1][0] == res
        if res:
            resamplings += 1
            np.testing.assert_array_equal(plans[0][1], idx)
            np.testing.assert_array_equal(plans[1][1], idx)
            blobs = np.concatenate([s.export() for s in shards])
            # exports first (every rank), imports afterwards
            needed = [{int(j) for j in idx[r * half:(r + 1) * half] if not (r * half <= j < (r + 1) * half)}
                      for r in range(2)]
            exported = {j: shards[j // half].export_particle_map(j % half) for j in set().union(*needed)}
            migrations += len(exported)
            for r, s in enumerate(shards):
                s.import_maps(blobs, idx, {j: exported[j] for j in needed[r]})
        poses, wts, ms = whole.state()
        sp = [s.state() for s in shards]
        np.testing.assert_array_equal(np.concatenate([x[0] for x in sp]), poses)
        np.testing.assert_array_equal(np.concatenate([x[1] for x in sp]), wts)
        np.testing.assert_array_equal(np.concatenate([x[2] for x in sp]), ms)
        if res or it % 6 == 0 or it == len(steps) - 1:
            for i in range(n):
                a_p, a_a = whole.particle_map(i, -ox, -oy, w, h)
                b_p, b_a = shards[i // half].particle_map(i % half, -ox, -oy, w, h)
                np.testing.assert_array_equal(b_p, a_p, err_msg="step %d particle %d" % (it, i))
                np.testing.assert_array_equal(b_a, a_a)
    assert resamplings >= 2 and migrations >= 1


def test_particle_maps_argument_checks(pkg):
    ctx = pkg.Context(0)
    ctx.map_bind(1, 2, 256, 256, (128, 128), 0.05, [0.5, 0, 0])
    pf = pkg.GmappingFilter(ctx, pkg.gmapping_params(), 8, np.arange(4, dtype=np.uint32), first=0, count=4)
    pf.enable_particle_maps(1, 4, 64)  # a shard holds the maps of its own particles
    with pytest.raises(pkg.SlamHipError):  # ... and resamples through import_maps, not import_
        pf.import_(np.zeros(8 * pf.blob_size(), np.uint8), np.arange(8, dtype=np.uint32))
    whole = pkg.GmappingFilter(ctx, pkg.gmapping_params(), 4, np.arange(4, dtype=np.uint32))
    with pytest.raises(pkg.SlamHipError):  # window larger than the tile extent
        whole.enable_particle_maps(1, 1, 64)
    whole.enable_particle_maps(1, 4, 64)
    with pytest.raises(pkg.SlamHipError):  # shared-map mode and particle maps exclude each other
        whole.set_map_update(True)


def test_particle_maps_vs_reference_copy_on_write_copies(pkg):
    """VERDICT r1 items 6 + 8: the tile pool against REAL copies of the compiled reference's
    UnboundedLazyTiledGridMap (tests/golden/make_golden_particle_maps.py): six copies of an ancestor map take
    one scan each from their own pose -- one batched K6 here -- with the AreaOccupancyEstimator and blur 0.1 m
    (BASELINE configs[4]); a copy of one of them moves on alone; originals must not see their copies' writes
    (lazy_tiled_grid_map.h:40-71)."""
    g = load("particle_maps_cow.npz")
    w, h = [int(v) for v in g["size"]]
    scale, blur, shift = float(g["scale"]), float(g["blur"]), float(g["shift_amount"])
    ox, oy = [int(v) for v in g["origin"]]
    base = tuple(g["base"])
    ctx = pkg.Context(0)
    ctx.map_bind(4, 2, w, h, g["origin"], scale, g["unknown"][:3])
    c0, s0 = pkg.beam_trig(g["scan0_angle"])
    ctx.map_append_scan(4, pkg.RULE_GMAPPING, g["pose0"], g["scan0_range"], c0, s0, is_occ=g["scan0_occ"], base=base,
                        blur=blur, estimator=1, shift_amount=shift)
    n = 8
    pf = pkg.GmappingFilter(ctx, pkg.gmapping_params(), n, np.arange(n, dtype=np.uint32))
    pf.enable_particle_maps(4, extent_tiles=8, pool_tiles=16 + 24 * n, base=base, blur=blur, estimator=1,
                            shift_amount=shift)

    def check(particle, name):
        got_p, got_a = pf.particle_map(particle, -ox, -oy, w, h)
        np.testing.assert_array_equal(got_a, g[name + "_aux"], err_msg=name)            # hits / tries exact
        np.testing.assert_allclose(got_p, g[name + "_payload"], rtol=1e-10, atol=1e-13, err_msg=name)

    for i in range(n):
        check(i, "A")  # every particle starts as the ancestor (one shared set of tiles)
    st0 = pf.particle_map_stats()
    nu = pf.particle_maps_append(np.arange(6), g["poses_b"], g["scan1_range"], g["scan1_angle"], g["scan1_occ"])
    assert nu > 6 * 360 * 20
    for i in range(6):
        check(i, "B%d" % i)
    check(6, "A")  # untouched particles still read the ancestor
    check(7, "A")
    st1 = pf.particle_map_stats()
    assert st1["cow_copies"] > st0["cow_copies"] and st1["tiles_shared"] > 0
    # `*new_particle = *sampled` (particle_filter.h:92-96): particle 6 becomes a copy of particle 2 -- a table
    # copy, no tile moves -- and then writes alone
    idx = np.array([0, 1, 2, 3, 4, 5, 2, 7], dtype=np.uint32)
    pf.import_(pf.export(), idx)
    st2 = pf.particle_map_stats()
    assert st2["cow_copies"] == st1["cow_copies"]
    check(6, "B2")
    pf.particle_maps_append([6], g["pose_c"], g["scan2_range"], g["scan2_angle"], g["scan2_occ"])
    check(6, "C")
    check(2, "B2")  # the original did not see the copy's writes
    check(7, "A")
    assert pf.particle_map_stats()["cow_copies"] > st2["cow_copies"]
    ctx.close()


def test_cfg5_geometry_cached_provider_golden(pkg):
    """cfg5's geometry against the compiled reference with no caveat (tests/golden/make_golden_cfg5_cached.py): 0.025 m
    cells, 1080-beam scans whose walks are up to 676 cells long, AreaOccupancyEstimator, blur 0.1 m, three scans per
    history -- with the reference's CACHED trigonometry provider, whose angle-addition form is the device's own
    arithmetic.  Every cell of every snapshot: hit / try counters exact, occupancy bit-exact, obstacle means to 1e-12
    (running means of end points).  Twice: the batched update of two particles' copy-on-write maps (histories P and Q
    in the same launches), and the single-scan update of a dense map (history P)."""
    from helpers import dense_snapshot
    g = load("cfg5_cached.npz")
    w, h = [int(v) for v in g["size"]]
    scale, blur, shift = float(g["scale"]), float(g["blur"]), float(g["shift_amount"])
    ox, oy = [int(v) for v in g["origin"]]
    base = tuple(g["base"])
    cos_t, sin_t = pkg.beam_trig(g["angle"])  # libm at the table's angles = the provider's table (asserted by the generator)

    def check(got_p, got_a, name):
        want_p, want_a = dense_snapshot(g, name)
        np.testing.assert_array_equal(got_a, want_a, err_msg=name)
        np.testing.assert_array_equal(got_p[..., 0], want_p[..., 0], err_msg=name)
        np.testing.assert_allclose(got_p[..., 1:], want_p[..., 1:], rtol=1e-12, atol=1e-14, err_msg=name)

    ctx = pkg.Context(0)
    ctx.map_bind(4, 2, w, h, g["origin"], scale, g["unknown"][:3])  # never observed: the particles' common ancestor
    pf = pkg.GmappingFilter(ctx, pkg.gmapping_params(), 2, np.arange(2, dtype=np.uint32))
    tiles = (max(w, h) + 127) // 128 + 1
    pf.enable_particle_maps(4, extent_tiles=tiles, pool_tiles=3 * tiles * tiles, base=base, blur=blur, estimator=1,
                            shift_amount=shift)
    ctx.map_bind(5, 2, w, h, g["origin"], scale, g["unknown"][:3])
    for k in range(3):
        rng, beam = g["scan%d_range" % k], g["scan%d_beam" % k]
        nu = pf.particle_maps_append([0, 1], np.stack([g["poses_p"][k], g["poses_q"][k]]), rng, g["angle"][beam])
        assert nu > 2 * rng.size * 100
        ctx.map_append_scan(5, pkg.RULE_GMAPPING, g["poses_p"][k], rng, cos_t[beam], sin_t[beam], base=base, blur=blur,
                            estimator=1, shift_amount=shift)
        if k == 0:
            check(*pf.particle_map(0, -ox, -oy, w, h), "P0")
            check(ctx.map_download_window(5, 0, 0, w, h, 3), ctx.map_download_aux(5, 0, 0, w, h, 2), "P0")
    check(*pf.particle_map(0, -ox, -oy, w, h), "P2")
    check(*pf.particle_map(1, -ox, -oy, w, h), "Q2")
    check(ctx.map_download_window(5, 0, 0, w, h, 3), ctx.map_download_aux(5, 0, 0, w, h, 2), "P2")
    pf.close()
    ctx.close()


@pytest.mark.parametrize("pose_trig", [1, 0])
def test_cfg5_geometry_against_the_oracle(pkg, oracle, pose_trig):
    """pose_trig 1: host pose trigonometry, host-driven lock-step jobs; 0 (the default, what bench.py runs): device
    sincos -- at six particles one accept chain per particle on the device, gathering through the particle's tile table
    (the lock-step jobs of larger shards have test_cfg5_matching_through_tile_tables_beyond_64_particles below).
    BASELINE configs[4] at its own geometry: an 8000 x 8000 map at 0.025 m per cell, 1080 beams that walk up to
    1200 cells, AreaOccupancyEstimator, blur 0.1 m (four cells), per-particle copy-on-write maps, the map update
    fused behind the likelihood.  Six particles instead of five hundred (the oracle keeps a dense map per particle),
    two steps: poses, weights and resampling decisions equal the oracle's, and EVERY particle's map over the whole
    world window -- payload to 1e-10, hit / try counters exact -- except where the documented raw-provider caveat
    bites: the oracle's filter, like the reference, takes libm sin(theta + a) for an end point,
    the device the angle-addition form; the last ulp is harmless for the const estimator, but the area estimator's
    are_on_the_same_side turns it into another area split for an occasional beam that grazes a cell corner (and
    with it the blurred cells in front of it).  Such cells must be EXPLAINED: the oracle itself, run beam by beam
    with the raw and with the device's trigonometry, has to differ in exactly those cells.  The GPU side binds the
    full 8000^2 ancestor; the oracle works on the 3200^2 window the synthetic world covers, same coordinates."""
    import pyoracle as po
    from pyoracle_mapupdate import RULE_GMAPPING, append_scan_ex, gmapping_enable_particle_maps, gmapping_particle_map
    from synth import make_scene
    size, win, scale, n = 8000, 3200, 0.025, 6
    sc = make_scene(cell_model=2, size=win, scale=scale, n_beams=1080, seed=6, blur_m=0.1)
    m, scan = sc["map"], sc["scan"]
    off = (size - win) // 2
    ctx = pkg.Context(0)
    ctx.map_bind(2, 2, size, size, (m.origin[0] + off, m.origin[1] + off), scale, m.unknown)
    ctx.map_upload_window(2, off, off, m.payload)
    gp = [0.0, 0.05, 0.0, 0.03, 0.0, 0.0, 0.0, 0.0]
    seeds = np.arange(3000, 3000 + n, dtype=np.uint32)
    shift = 0.01 * scale
    pf = pkg.GmappingFilter(ctx, pkg.gmapping_params(gp8=gp, pose_trig=pose_trig), n, seeds)
    ext = (size + 127) // 128 + 1
    reach = int(np.ceil(2.0 * (float(scan.range.max()) + 1.0) / scale / 128.0)) + 2
    pf.enable_particle_maps(2, extent_tiles=ext, pool_tiles=ext * ext + n * reach * reach, blur=0.1, estimator=1,
                            shift_amount=shift)
    aux = np.zeros((win, win, 2))
    opf = oracle.gmapping_create(n, gp, seeds)
    gmapping_enable_particle_maps(oracle, opf, m, aux, blur=0.1, est_kind=1, shift_amount=shift)
    rs = np.random.RandomState(8)
    deltas = [sc["true_pose"], rs.randn(3) * [0.03, 0.03, 0.01]]
    ranges = np.clip(scan.range + rs.randn(scan.n) * 0.01, 0.05, None)  # (N(0, 0.01 m) range noise)
    ox, oy = m.origin
    poses_at = []
    for k, d in enumerate(deltas):
        extra = np.arange(9000 + 100 * k, 9000 + 100 * k + n, dtype=np.uint32)
        res, idx = pf.step(2, ranges, scan.angle, None, d, 7 + k)
        ores, oidx = opf.step(m, ranges, scan.angle, None, d, 7 + k, extra)
        poses, wts, ms = pf.state()
        oposes, owts, oms = opf.state()
        assert res == ores
        if res:
            np.testing.assert_array_equal(idx, oidx)
        np.testing.assert_allclose(poses, oposes, rtol=0, atol=1e-10, err_msg="step %d" % k)
        np.testing.assert_allclose(wts, owts, rtol=1e-9, atol=0, err_msg="step %d" % k)
        poses_at.append(poses.copy())
    assert pf.stats()["scorer_calls"] == opf.o.lib.orc_gmapping_scorer_calls(opf.h)
    assert pf.particle_map_stats()["cell_updates"] > n * 1080 * 300  # long beams: hundreds of cells each
    c_all, s_all = pkg.beam_trig(scan.angle)
    unexplained, caveat_cells = 0, 0
    for i in range(n):
        got_p, got_a = pf.particle_map(i, -ox, -oy, win, win)
        want_p, want_a = gmapping_particle_map(oracle, opf, i)
        bad = (got_a != want_a).any(-1) | ~np.isclose(got_p, want_p, rtol=1e-10, atol=1e-13).all(-1)
        if not bad.any():
            continue
        cells = np.argwhere(bad)  # (row, col) in the window
        explained = np.zeros_like(bad)
        small, half = 64, 32
        for k in range(len(deltas)):
            px, py, pth = poses_at[k][i]
            ex = (px + ranges * np.cos(pth + scan.angle)) / scale
            ey = (py + ranges * np.sin(pth + scan.angle)) / scale
            near = np.zeros(scan.n, bool)
            for (yy, xx) in cells:
                near |= (np.abs(ex - (xx - ox)) < 8) & (np.abs(ey - (yy - oy)) < 8)
            for b in np.nonzero(near)[0]:
                # one beam into an empty scratch window around its end point, raw against device trigonometry
                cx0, cy0 = int(np.floor(ex[b])) - half, int(np.floor(ey[b])) - half
                r1, a1 = ranges[b:b + 1].copy(), scan.angle[b:b + 1].copy()
                outs = []
                for mode in ("raw", "device"):
                    pay = np.tile(np.asarray(m.unknown, dtype=np.float64)[:3], (small, small, 1))
                    sm = po.GridMapData(po.CELL_GMAPPING, pay, (-cx0, -cy0), scale, m.unknown)
                    sa = np.zeros((small, small, 2))
                    # only the last cells of the beam matter: start it 1 m before its end
                    cut = max(r1[0] - 1.0, 0.0)
                    start = np.array([px + cut * np.cos(pth + a1[0]), py + cut * np.sin(pth + a1[0]), pth])
                    rr = r1 - cut
                    if mode == "raw":
                        append_scan_ex(oracle, sm, sa, RULE_GMAPPING, start, rr, a1, None, blur=0.1, est_kind=1, shift_amount=shift)
                    else:
                        tr = po.ScanData(rr, a1, None, None, po.TRIG_CACHED, 0.0, 1.0, s_all[b:b + 1].copy(), c_all[b:b + 1].copy())
                        tr.angle = np.arange(1, dtype=np.float64)
                        append_scan_ex(oracle, sm, sa, RULE_GMAPPING, start, rr, tr.angle, None, blur=0.1, est_kind=1,
                                       shift_amount=shift, trig=tr)
                    outs.append((pay, sa))
                diff = (outs[0][1] != outs[1][1]).any(-1) | ~np.isclose(outs[0][0], outs[1][0], rtol=1e-10, atol=1e-13).all(-1)
                for (yy, xx) in np.argwhere(diff):
                    gy, gx = yy + cy0 + oy, xx + cx0 + ox
                    if 0 <= gy < win and 0 <= gx < win:
                        explained[gy, gx] = True
        caveat_cells += int(bad.sum())
        unexplained += int((bad & ~explained).sum())
    print("cfg5 geometry: %d cells under the raw-provider caveat, %d unexplained" % (caveat_cells, unexplained))
    assert unexplained == 0, "%d map cells differ from the oracle outside the raw-provider caveat" % unexplained
    assert caveat_cells <= 64, "the caveat is rare: %d cells" % caveat_cells
    pf.close()
    ctx.close()


@pytest.mark.parametrize("case", ["fresh_map", "area_blur", "negative_blur", "free_points_and_range_gate",
                                  "occupied_base_empty"])
def test_fast_path_equals_the_sorted_chains(pkg, case):
    """The batched update's free-space fast path (k_mu_classify: atomics on the try counters of zero-mean and
    never-observed cells) against the pipeline that sorts every record into its cell's chain, on the same pool
    contents, bit for bit over every particle's whole map -- three scans in a row, so the second and third meet
    cells the first one created.  Cases the oracle runs do not reach: a map nobody has written to, a blur that
    scales with the beam length, points flagged free plus a range gate with non-finite ranges behind it, and a
    base_empty probability above 0.5 (a "free" observation is then a hit: the fast path must stand aside)."""
    from synth import make_scene
    n, size, scale = 5, 512, 0.05
    sc = make_scene(cell_model=2, size=size, scale=scale, n_beams=360, seed=11)
    m, scan = sc["map"], sc["scan"]
    rs = np.random.RandomState(3)
    poses = sc["true_pose"] + rs.randn(n, 3) * [0.05, 0.05, 0.02]
    adder = {}
    occ = None
    rng = scan.range.copy()
    if case == "area_blur":
        adder = {"blur": 0.25, "estimator": 1, "shift_amount": 0.01 * scale}
    if case == "negative_blur":
        adder = {"blur": -0.02}
    if case == "free_points_and_range_gate":
        adder = {"max_range": float(np.percentile(rng, 70)), "blur": 0.15}
        occ = (rs.rand(scan.n) < 0.7).astype(np.int32)
        rng[::17] = np.inf
        rng[5::23] = 1e9
    if case == "occupied_base_empty":
        adder = {"base": (0.95, 1.0, 0.6, 1.0)}
    maps = {}
    for mode in ("fast", "sorted"):
        ctx = pkg.Context(0)
        ctx.set_option(pkg.OPT_K6_BATCH_FAST, 1 if mode == "fast" else 0)
        ctx.map_bind(3, 2, size, size, m.origin, scale, m.unknown)
        if case != "fresh_map":
            ctx.map_upload_window(3, 0, 0, m.payload)
        pf = pkg.GmappingFilter(ctx, pkg.gmapping_params(), n, np.arange(n, dtype=np.uint32))
        pf.enable_particle_maps(3, extent_tiles=6, pool_tiles=8 + 20 * n, **adder)
        total = 0
        for k in range(3):
            total += pf.particle_maps_append(np.arange(n), poses + 0.01 * k, rng, scan.angle, occ)
        ox, oy = m.origin
        maps[mode] = (total, [pf.particle_map(i, -ox, -oy, size, size) for i in range(n)])
        pf.close()
        ctx.close()
    assert maps["fast"][0] == maps["sorted"][0] > 0
    touched = 0
    for (fp, fa), (sp, sa) in zip(maps["fast"][1], maps["sorted"][1]):
        assert fp.tobytes() == sp.tobytes()
        assert fa.tobytes() == sa.tobytes()
        touched += int(np.count_nonzero(fa[..., 1]))
    assert touched > n * 1000


def test_fast_path_at_full_batch_size_counts_every_record_once(pkg):
    """The 100-particle batch of BASELINE configs[3] (1080 beams, 0.05 m cells) through both pipelines: the number of
    cell updates, the sum of all try counters of a particle (every valid record is exactly one try, whichever path
    settled it) and the whole map of sampled particles, byte for byte.  Two batches, so that the second meets the
    hits of the first."""
    from synth import make_scene
    n, size, scale = 100, 2000, 0.05
    sc = make_scene(cell_model=2, size=size, scale=scale, n_beams=1080, seed=21)
    m, scan = sc["map"], sc["scan"]
    rs = np.random.RandomState(5)
    poses = sc["true_pose"] + rs.randn(n, 3) * [0.1, 0.1, 0.03]
    sample = [0, 37, 99]
    out = {}
    for mode in ("fast", "sorted"):
        ctx = pkg.Context(0)
        ctx.set_option(pkg.OPT_K6_BATCH_FAST, 1 if mode == "fast" else 0)
        ctx.map_bind(3, 2, size, size, m.origin, scale, m.unknown)
        ctx.map_upload_window(3, 0, 0, m.payload)
        pf = pkg.GmappingFilter(ctx, pkg.gmapping_params(), n, np.arange(n, dtype=np.uint32))
        pf.enable_particle_maps(3, extent_tiles=(size + 127) // 128 + 1, pool_tiles=300 + 150 * n)
        nu = [pf.particle_maps_append(np.arange(n), poses + 0.02 * k, scan.range, scan.angle) for k in range(2)]
        ox, oy = m.origin
        maps = [pf.particle_map(i, -ox, -oy, size, size) for i in sample]
        out[mode] = (nu, maps)
        pf.close()
        ctx.close()
    assert out["fast"][0] == out["sorted"][0]
    assert min(out["fast"][0]) > n * 1080 * 50
    for (fp, fa), (sp, sa) in zip(out["fast"][1], out["sorted"][1]):
        assert fp.tobytes() == sp.tobytes() and fa.tobytes() == sa.tobytes()
    # the ancestor carries no counters: a particle's tries are its records of the two batches
    tries = [float(fa[..., 1].sum()) for fp, fa in out["fast"][1]]
    assert abs(np.mean(tries) * n - sum(out["fast"][0])) < 0.05 * sum(out["fast"][0])


def _device_trig_scan(pkg, po, rng, ang):
    """The scan as the device sees it: per-beam libm cos / sin of the scan angle (slamhip_beam_trig_raw), combined
    with the pose heading by angle addition -- for the oracle a cached-provider scan whose table slot i holds beam i."""
    cos_a, sin_a = pkg.beam_trig(ang)
    tr = po.ScanData(rng, ang, trig_mode=po.TRIG_CACHED, a_min=0.0, a_delta=1.0, tab_sin=sin_a, tab_cos=cos_a)
    tr.angle = np.arange(len(ang), dtype=np.float64)
    return tr


@pytest.mark.parametrize("n", [70, 120])
def test_cfg5_matching_through_tile_tables_beyond_64_particles(pkg, oracle, n):
    """What bench.py's cfg5 and pf_maps legs run for their likelihood step, against the oracle: MORE than 64 particles
    with per-particle maps and the default device pose trigonometry.  120 particles match through host-driven
    lock-step jobs whose K3 launches resolve every gather through the particle's tile table (cfg5's path); 70 fit ONE
    co-resident launch of per-particle chains (csrc/gmapping.cpp: gm_multi_chain_fits_resident, r04 -- before, chains
    only up to 64), which gather through the same tables.  Cell
    size 0.025 m, area estimator, blur 0.1 m.  So that every particle reads a DIFFERENT map, each one first takes the
    scan from its own jittered pose (one batched K6 here; the oracle appends with the device's trigonometry form, so
    the raw-provider caveat of the cfg5 geometry test cannot arise) -- sampled maps are compared bit for bit on the
    counters -- and then ONE filter step runs on both sides: poses 1e-10, weights 1e-9, scorer calls and resampling
    decision exact (gmapping_world.h:73-101)."""
    import pyoracle as po
    from pyoracle_mapupdate import (gmapping_enable_particle_maps, gmapping_particle_map, gmapping_particle_map_append)
    from synth import make_scene
    win, scale = 1280, 0.025
    sc = make_scene(cell_model=2, size=win, scale=scale, n_beams=1080, seed=9, blur_m=0.1, max_dist=12.0)
    m, scan = sc["map"], sc["scan"]
    ctx = pkg.Context(0)
    ctx.map_bind(2, 2, win, win, m.origin, scale, m.unknown)
    ctx.map_upload_window(2, 0, 0, m.payload)
    gp = [0.0, 0.05, 0.0, 0.03, 0.0, 0.0, 0.0, 0.0]
    seeds = np.arange(4000, 4000 + n, dtype=np.uint32)
    shift = 0.01 * scale
    pf = pkg.GmappingFilter(ctx, pkg.gmapping_params(gp8=gp, pose_trig=0), n, seeds)
    ext = (win + 127) // 128 + 1
    pf.enable_particle_maps(2, extent_tiles=ext, pool_tiles=ext * ext + n * ext * ext, blur=0.1, estimator=1,
                            shift_amount=shift)
    aux = np.zeros((win, win, 2))
    opf = oracle.gmapping_create(n, gp, seeds)
    gmapping_enable_particle_maps(oracle, opf, m, aux, blur=0.1, est_kind=1, shift_amount=shift)
    rs = np.random.RandomState(12)
    poses0 = sc["true_pose"] + rs.randn(n, 3) * [0.04, 0.04, 0.01]
    nu = pf.particle_maps_append(np.arange(n), poses0, scan.range, scan.angle)
    tr = _device_trig_scan(pkg, po, scan.range, scan.angle)
    onu = sum(gmapping_particle_map_append(oracle, opf, m, i, poses0[i], scan.range, tr.angle, None, trig=tr)
              for i in range(n))
    assert nu == onu > n * 1080 * 100
    ox, oy = m.origin
    for i in (0, 33, n - 1):
        got_p, got_a = pf.particle_map(i, -ox, -oy, win, win)
        want_p, want_a = gmapping_particle_map(oracle, opf, i)
        np.testing.assert_array_equal(got_a, want_a, err_msg="particle %d" % i)
        np.testing.assert_allclose(got_p, want_p, rtol=1e-10, atol=1e-13)
    extra = np.arange(9000, 9000 + n, dtype=np.uint32)
    res, idx = pf.step(2, scan.range, scan.angle, None, sc["true_pose"], 7)
    ores, oidx = opf.step(m, scan.range, scan.angle, None, sc["true_pose"], 7, extra)
    poses, wts, ms = pf.state()
    oposes, owts, oms = opf.state()
    assert res == ores
    if res:
        np.testing.assert_array_equal(idx, oidx)
    np.testing.assert_array_equal(ms, oms)
    np.testing.assert_allclose(poses, oposes, rtol=0, atol=1e-10)
    np.testing.assert_allclose(wts, owts, rtol=1e-9, atol=0)
    assert pf.stats()["scorer_calls"] == opf.o.lib.orc_gmapping_scorer_calls(opf.h)
    assert len(np.unique(np.round(wts, 12))) > n // 2  # the particles really read different maps
    pf.close()
    ctx.close()


def _cfg5_batch(pkg, sc, poses, particles_per_call, sample, fast, size=8000):
    """One particle_maps_append per group of `particles_per_call` particles at cfg5's geometry; returns the number of
    cell updates and, per sampled particle, (sha256 of its payload window, sha256 of its counters, sum of tries)."""
    import hashlib
    m, scan = sc["map"], sc["scan"]
    win, scale = m.width, m.scale
    off = (size - win) // 2
    out, total = {}, 0
    n = len(poses)
    for g0 in range(0, n, particles_per_call):
        grp = np.arange(g0, min(n, g0 + particles_per_call))
        if particles_per_call < n and not any(int(i) in sample for i in grp):
            continue  # a group no sampled particle belongs to says nothing in the small-group runs
        ctx = pkg.Context(0)
        ctx.set_option(pkg.OPT_K6_BATCH_FAST, 1 if fast else 0)
        ctx.map_bind(2, 2, size, size, (m.origin[0] + off, m.origin[1] + off), scale, m.unknown)
        ctx.map_upload_window(2, off, off, m.payload)
        k = len(grp)
        pf = pkg.GmappingFilter(ctx, pkg.gmapping_params(), k, np.arange(k, dtype=np.uint32))
        ext = (size + 127) // 128 + 1
        reach = int(np.ceil(2.0 * (float(scan.range.max()) + 1.0) / scale / 128.0)) + 2
        pf.enable_particle_maps(2, extent_tiles=ext, pool_tiles=ext * ext + k * reach * reach, blur=0.1, estimator=1,
                                shift_amount=0.01 * scale)
        total += pf.particle_maps_append(np.arange(k), poses[grp], scan.range, scan.angle)
        ox, oy = m.origin
        for j, i in enumerate(grp):
            if int(i) in sample:
                pay, aux = pf.particle_map(j, -ox, -oy, win, win)
                out[int(i)] = (hashlib.sha256(pay.tobytes()).hexdigest(), hashlib.sha256(aux.tobytes()).hexdigest(),
                               float(aux[..., 1].sum()))
        pf.close()
        ctx.close()
    return total, out


def test_cfg5_batch_of_500_particles_by_its_properties(pkg):
    """BASELINE configs[4] at FULL batch size -- 500 particles, 8000 x 8000 cells of 0.025 m, 1080 beams, area
    estimator, blur 0.1 m, ONE batched K6 (190 M records) -- where the oracle cannot follow (a dense map per
    particle).  Size-independent properties instead: (1) a particle's map does not depend on who else is in the batch:
    sampled particles' maps equal, byte for byte, the maps the same particles get in batches of six (the size
    test_cfg5_geometry_against_the_oracle pins to the oracle); (2) the free-space fast path (atomics + sorted rest) and
    the pipeline that sorts every record agree byte for byte at the full batch size; (3) every valid record is
    exactly one try: a particle's try counters add up to its records."""
    from synth import make_scene
    n, win, scale = 500, 3200, 0.025
    sc = make_scene(cell_model=2, size=win, scale=scale, n_beams=1080, seed=6, blur_m=0.1)
    rs = np.random.RandomState(17)
    poses = sc["true_pose"] + rs.randn(n, 3) * [0.05, 0.05, 0.01]
    sample = {0, 5, 131, 250, 377, 499}
    total_fast, fast = _cfg5_batch(pkg, sc, poses, n, sample, True)
    total_sorted, srt = _cfg5_batch(pkg, sc, poses, n, sample, False)
    _, small = _cfg5_batch(pkg, sc, poses, 6, sample, True)
    assert total_fast == total_sorted > n * 1080 * 300
    assert set(fast) == set(srt) == set(small) == sample
    for i in sorted(sample):
        assert fast[i][:2] == srt[i][:2], "particle %d: fast path != sorted chains at 500 particles" % i
        assert fast[i][:2] == small[i][:2], "particle %d: its map depends on the batch it was in" % i
    assert len({v[1] for v in fast.values()}) == len(sample)  # different poses, different maps
    tries = np.array([v[2] for v in fast.values()])
    assert abs(tries.mean() * n - total_fast) < 0.02 * total_fast


def test_cfg5_batch_of_500_direct_oracle_replay_of_sampled_particles(pkg, oracle):
    """the 500-particle check above is transitive (500-batch = 6-batch = oracle).  With per-particle
    maps a particle's map depends on its OWN pose history only (gmapping_world.h:93-97), so the oracle can replay a
    sample of the 500 directly: three batched K6 appends of all 500 particles at cfg5's geometry (8000 x 8000 cells of
    0.025 m through tile tables, area estimator, blur 0.1 m) while the oracle appends the same three scans to eight
    sampled particles' dense maps from the same poses -- counters bit for bit, running means to 1e-10, update counts
    equal (area_occupancy_estimator.h:27-240, grid_map_scan_adders.h:54-75,138-172, gmapping_grid_cell.h:20-33)."""
    import pyoracle as po
    from pyoracle_mapupdate import (gmapping_enable_particle_maps, gmapping_particle_map, gmapping_particle_map_append)
    from synth import make_scene
    n, win, scale, size = 500, 1280, 0.025, 8000
    sc = make_scene(cell_model=2, size=win, scale=scale, n_beams=1080, seed=9, blur_m=0.1, max_dist=12.0)
    m, scan = sc["map"], sc["scan"]
    off = (size - win) // 2
    shift = 0.01 * scale
    sample = [0, 7, 131, 250, 251, 377, 498, 499]
    rs = np.random.RandomState(23)
    hist = [sc["true_pose"] + rs.randn(n, 3) * [0.05, 0.05, 0.01]]
    for _ in range(2):
        hist.append(hist[-1] + rs.randn(n, 3) * [0.03, 0.03, 0.008])
    ctx = pkg.Context(0)
    ctx.map_bind(2, 2, size, size, (m.origin[0] + off, m.origin[1] + off), scale, m.unknown)
    ctx.map_upload_window(2, off, off, m.payload)
    pf = pkg.GmappingFilter(ctx, pkg.gmapping_params(), n, np.arange(n, dtype=np.uint32))
    ext = (size + 127) // 128 + 1
    reach = int(np.ceil(2.0 * (float(scan.range.max()) + 1.0) / scale / 128.0)) + 2
    pf.enable_particle_maps(2, extent_tiles=ext, pool_tiles=ext * ext + n * reach * reach, blur=0.1, estimator=1,
                            shift_amount=shift)
    # the oracle's filter holds only the sampled particles, each on its own dense copy of the window
    k = len(sample)
    opf = oracle.gmapping_create(k, [0.0] * 8, np.arange(k, dtype=np.uint32))
    gmapping_enable_particle_maps(oracle, opf, m, np.zeros((win, win, 2)), blur=0.1, est_kind=1, shift_amount=shift)
    tr = _device_trig_scan(pkg, po, scan.range, scan.angle)
    for step, poses in enumerate(hist):
        total = pf.particle_maps_append(np.arange(n), poses, scan.range, scan.angle)
        assert total > n * 1080 * 100
        want_updates = sum(gmapping_particle_map_append(oracle, opf, m, j, poses[i], scan.range, tr.angle, None, trig=tr)
                           for j, i in enumerate(sample))
        # (the batch reports one number for all 500; the sample's share is checked through the try counters below)
        assert want_updates > k * 1080 * 100
    ox, oy = m.origin
    for j, i in enumerate(sample):
        got_p, got_a = pf.particle_map(i, -ox, -oy, win, win)
        want_p, want_a = gmapping_particle_map(oracle, opf, j)
        np.testing.assert_array_equal(got_a, want_a, err_msg="particle %d: hit / try counters after three scans" % i)
        np.testing.assert_allclose(got_p, want_p, rtol=1e-10, atol=1e-13, err_msg="particle %d" % i)
        assert got_a[..., 1].sum() > 3 * 1080 * 100
    pf.close()
    ctx.close()
