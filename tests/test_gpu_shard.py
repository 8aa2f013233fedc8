"""GPU suite: the sharded GMapping step behind the C-ABI (csrc/shard.cpp, slamhip_gmapping_step_sharded).

  * one rank: RCCL really runs (ncclCommInitRank / ncclAllGather with world = 1) and the sharded step equals
    the unsharded one bit for bit;
  * two shards on ONE GPU (RCCL refuses two ranks on one device, so the records travel through Python here):
    the phases match_begin / carry_record / carry_fix / match_finish against the unsharded filter, also in a
    scene built to make the shared OOPE cache of the reference (Q19/Q20) hit ACROSS the shard boundary;
  * two ranks on two GPUs over RCCL (skipped on a 1-GPU box)."""
import os
import sys

import ctypes as C

import numpy as np
import pytest
from synth import make_scene

import __graft_entry__ as ge

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GP = [0.0, 0.1, 0.0, 0.03, 0.0, 0.0, 0.0, 0.0]  # gate open: every particle matches


@pytest.fixture(scope="module")
def pkg():
    return ge.load_package()


def deltas(sc, n):
    # large odometry steps open the travelled-distance gate of try_resample and spread the weights
    d = [sc["true_pose"], [0.02, 0.01, 0.01], [0.4, 0.5, 0.3], [0.01, -0.02, 0.02], [0.5, -0.4, 0.25],
         [0.02, 0.02, 0.0], [0.45, 0.5, -0.3], [0.0, 0.01, 0.01]]
    return d[:n]


def test_one_rank_over_rccl_equals_unsharded_step(pkg):
    ctx = pkg.Context(0)
    ctx.shard_init(0, 1, pkg.shard_unique_id())
    assert ctx.shard_info() == (0, 1)
    blk = np.arange(12, dtype=np.float64).reshape(4, 3)
    np.testing.assert_array_equal(ctx.shard_allgather(blk, [4]), blk)
    sc = make_scene(cell_model=2, size=800, scale=0.05, n_beams=360, seed=4)
    ctx.upload_map(1, sc["map"])
    n = 24
    seeds = np.arange(1000, 1000 + n, dtype=np.uint32)
    a = pkg.GmappingFilter(ctx, pkg.gmapping_params(gp8=GP), n, seeds)
    b = pkg.GmappingFilter(ctx, pkg.gmapping_params(gp8=GP), n, seeds)
    resampled = 0
    for k, d in enumerate(deltas(sc, 8)):
        ra, ia = a.step_sharded(1, sc["scan"].range, sc["scan"].angle, None, d, 7 + k)
        rb, ib = b.step(1, sc["scan"].range, sc["scan"].angle, None, d, 7 + k)
        assert ra == rb
        if ra:
            np.testing.assert_array_equal(ia, ib)
            resampled += 1
        for x, y in zip(a.state(), b.state()):
            np.testing.assert_array_equal(x, y)
    assert resampled >= 1
    st = ctx.shard_stats()
    # the all-gather above and the block sizes once, then ONE collective per step (carry records + raw weights)
    # and the particle records of every resampling
    assert st["collectives"] == 2 + 8 + resampled and st["bytes"] > 0
    ctx.shard_destroy()
    ctx.close()


def run_two_shards(pkg, ctx, sc, n, cut, steps, scan, gp=GP):
    """Unsharded filter vs shards [0, cut) and [cut, n) on one GPU; the collectives are numpy concatenations."""
    seeds = np.arange(2000, 2000 + n, dtype=np.uint32)
    prm = pkg.gmapping_params(gp8=gp)
    whole = pkg.GmappingFilter(ctx, prm, n, seeds)
    sh = [pkg.GmappingFilter(ctx, prm, n, seeds[:cut], first=0, count=cut),
          pkg.GmappingFilter(ctx, prm, n, seeds[cut:], first=cut, count=n - cut)]
    for s in sh:
        s.set_shard_chain(True)
    reruns_whole, reruns_sh, resampled = 0, 0, 0
    for k, d in enumerate(steps):
        rw, iw = whole.step(1, scan.range, scan.angle, None, d, 7 + k)
        reruns_whole += whole.stats()["carry_reruns"]
        for s in sh:
            s.match_begin(1, scan.range, scan.angle, None, d)
        for _ in range(3):
            recs = [s.carry_record() for s in sh]
            changed = [s.carry_fix(recs, r) for r, s in enumerate(sh)]
            if not any(changed):
                break
        recs = [s.carry_record() for s in sh]
        for s in sh:
            s.carry_commit(recs)
        raw = np.concatenate([s.match_finish() for s in sh])
        reruns_sh += sum(s.stats()["carry_reruns"] for s in sh)
        plans = [s.plan_resample(raw, 7 + k) for s in sh]
        assert plans[0][0] == plans[1][0] == rw
        if rw:
            resampled += 1
            np.testing.assert_array_equal(plans[0][1], iw)
            np.testing.assert_array_equal(plans[1][1], iw)
            blobs = np.concatenate([s.export() for s in sh])
            for s in sh:
                s.import_(blobs, iw)
        pw, ww, mw = whole.state()
        ps = np.concatenate([s.state()[0] for s in sh])
        ws = np.concatenate([s.state()[1] for s in sh])
        ms = np.concatenate([s.state()[2] for s in sh])
        np.testing.assert_array_equal(ps, pw)
        np.testing.assert_array_equal(ws, ww)
        np.testing.assert_array_equal(ms, mw)
    return reruns_whole, reruns_sh, resampled


def test_two_shards_in_phases_equal_the_unsharded_filter(pkg):
    ctx = pkg.Context(0)
    sc = make_scene(cell_model=2, size=800, scale=0.05, n_beams=360, seed=4)
    ctx.upload_map(1, sc["map"])
    _, _, resampled = run_two_shards(pkg, ctx, sc, 21, 11, deltas(sc, 8), sc["scan"])
    assert resampled >= 1
    ctx.close()


def test_cache_hit_across_the_shard_boundary(pkg):
    """A scan of very few beams and particles that all start from one pose: the first beam of a particle's
    first pose lands in the cell its predecessor's last beam ended in, with another cached value -- the
    reference's shared OOPE cache then changes the score (Q19/Q20).  Inside a shard verify_chain re-matches
    such a particle; across the boundary carry_fix has to."""
    from synth import Scan
    ctx = pkg.Context(0)
    sc = make_scene(cell_model=2, size=800, scale=0.05, n_beams=360, seed=4)
    ctx.upload_map(1, sc["map"])
    full = sc["scan"]
    hits_whole = hits_sh = 0
    for pick in ([10], [200, 201], [100]):
        scan = Scan(full.range[pick], full.angle[pick], np.full(len(pick), 1.0 / len(pick)))
        gp = [0.0, 1e-9, 0.0, 1e-9, 0.0, 0.0, 0.0, 0.0]  # (almost) no pose noise: particles stay together
        steps = [sc["true_pose"], [0.02, 0.01, 0.0], [0.0, 0.03, 0.01], [0.01, 0.0, 0.0]]
        w, s, _ = run_two_shards(pkg, ctx, sc, 9, 4, steps, scan, gp=gp)
        hits_whole += w
        hits_sh += s
    assert hits_whole > 0, "the scene does not exercise the shared cache at all"
    assert hits_sh > 0
    ctx.close()


def _rank_main(rank, world, uid_path, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import __graft_entry__ as g2
    from synth import make_scene as mk
    pkg = g2.load_package()
    import time
    if rank == 0:
        uid = pkg.shard_unique_id()
        np.save(uid_path + ".tmp.npy", uid)
        os.replace(uid_path + ".tmp.npy", uid_path)
    else:
        for _ in range(600):
            if os.path.exists(uid_path):
                break
            time.sleep(0.05)
        uid = np.load(uid_path)
    ctx = pkg.Context(rank)
    ctx.shard_init(rank, world, uid)
    sc = mk(cell_model=2, size=800, scale=0.05, n_beams=360, seed=4)
    ctx.upload_map(1, sc["map"])
    n = 21
    counts = [n // world + (1 if r < n % world else 0) for r in range(world)]
    first = sum(counts[:rank])
    seeds = np.arange(2000, 2000 + n, dtype=np.uint32)
    pf = pkg.GmappingFilter(ctx, pkg.gmapping_params(gp8=GP), n, seeds[first:first + counts[rank]], first=first,
                            count=counts[rank])
    log = []
    for k, d in enumerate(deltas(sc, 8)):
        res, idx = pf.step_sharded(1, sc["scan"].range, sc["scan"].angle, None, d, 7 + k)
        p, w, m = pf.state()
        log.append((res, idx.copy(), p, w, m))
    np.save(out_path % rank, np.array(log, dtype=object), allow_pickle=True)
    ctx.shard_destroy()
    ctx.close()


def test_two_ranks_over_rccl(pkg, tmp_path):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    import multiprocessing as mp
    mpc = mp.get_context("spawn")
    uid_path, out_path = str(tmp_path / "uid.npy"), str(tmp_path / "rank%d.npy")
    procs = [mpc.Process(target=_rank_main, args=(r, 2, uid_path, out_path)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    logs = [np.load(out_path % r, allow_pickle=True) for r in range(2)]
    ctx = pkg.Context(0)
    sc = make_scene(cell_model=2, size=800, scale=0.05, n_beams=360, seed=4)
    ctx.upload_map(1, sc["map"])
    n = 21
    whole = pkg.GmappingFilter(ctx, pkg.gmapping_params(gp8=GP), n, np.arange(2000, 2000 + n, dtype=np.uint32))
    for k, d in enumerate(deltas(sc, 8)):
        rw, iw = whole.step(1, sc["scan"].range, sc["scan"].angle, None, d, 7 + k)
        assert logs[0][k][0] == logs[1][k][0] == rw
        if rw:
            np.testing.assert_array_equal(logs[0][k][1], iw)
        pw, ww, mw = whole.state()
        np.testing.assert_array_equal(np.concatenate([logs[0][k][2], logs[1][k][2]]), pw)
        np.testing.assert_array_equal(np.concatenate([logs[0][k][3], logs[1][k][3]]), ww)
    ctx.close()


def run_loopback_ranks(pkg, world, n, scene, steps, scan, gp, name):
    """slamhip_gmapping_step_sharded on `world` ranks = threads, one context each on GPU 0; returns per-rank logs"""
    import threading
    import loopback
    loopback.lib()  # (built once, before the threads start)
    counts = [n // world + (1 if r < n % world else 0) for r in range(world)]
    seeds = np.arange(2000, 2000 + n, dtype=np.uint32)
    logs, errors = [None] * world, []

    def rank_main(rank):
        try:
            ctx = pkg.Context(0)
            loopback.attach(pkg, ctx, name, rank, world)
            ctx.upload_map(1, scene["map"])
            first = sum(counts[:rank])
            pf = pkg.GmappingFilter(ctx, pkg.gmapping_params(gp8=gp), n, seeds[first:first + counts[rank]], first=first,
                                    count=counts[rank])
            log = []
            for k, d in enumerate(steps):
                res, idx = pf.step_sharded(1, scan.range, scan.angle, None, d, 7 + k)
                p, w, m = pf.state()
                log.append((res, None if idx is None else np.array(idx).copy(), p, w, m, pf.stats()["carry_reruns"]))
            logs[rank] = (log, ctx.shard_stats())
            pf.close()
            ctx.shard_destroy()
            ctx.close()
        except Exception as e:  # noqa: BLE001
            errors.append((rank, repr(e)))

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    assert all(not t.is_alive() for t in threads), "a rank is stuck in a collective"
    return logs, counts


@pytest.mark.parametrize("world", [2, 3])
def test_step_sharded_over_an_in_process_group(pkg, world):
    """The library's own sharded step (slamhip_gmapping_step_sharded: match, ONE all-gather of carry records +
    raw weights, identical resampling everywhere, records all-gathered when a resampling happens) with world > 1
    on one GPU: the ranks are threads, the transport an in-process board handed in through slamhip_shard_attach
    (tests/native/loopback_transport.cpp; the RCCL communicator admits one rank per device).  Every rank must hold exactly the particles of the unsharded filter after every step."""
    ctx = pkg.Context(0)
    sc = make_scene(cell_model=2, size=800, scale=0.05, n_beams=360, seed=4)
    ctx.upload_map(1, sc["map"])
    n = 21
    steps = deltas(sc, 8)
    whole = pkg.GmappingFilter(ctx, pkg.gmapping_params(gp8=GP), n, np.arange(2000, 2000 + n, dtype=np.uint32))
    logs, counts = run_loopback_ranks(pkg, world, n, sc, steps, sc["scan"], GP, "steps-%d" % world)
    resampled = 0
    for k, d in enumerate(steps):
        rw, iw = whole.step(1, sc["scan"].range, sc["scan"].angle, None, d, 7 + k)
        pw, ww, mw = whole.state()
        resampled += int(rw)
        for r in range(world):
            res, idx, p, w, m, _ = logs[r][0][k]
            assert res == rw
            if rw:
                np.testing.assert_array_equal(idx, iw)
        np.testing.assert_array_equal(np.concatenate([logs[r][0][k][2] for r in range(world)]), pw)
        np.testing.assert_array_equal(np.concatenate([logs[r][0][k][3] for r in range(world)]), ww)
        np.testing.assert_array_equal(np.concatenate([logs[r][0][k][4] for r in range(world)]), mw)
    assert resampled >= 1
    # one collective per step (+ the block sizes once, + the particle records of every resampling)
    for r in range(world):
        assert logs[r][1]["collectives"] == 1 + len(steps) + resampled, logs[r][1]
    ctx.close()


def test_step_sharded_repairs_the_cache_across_ranks(pkg):
    """The scene of test_cache_hit_across_the_shard_boundary through slamhip_gmapping_step_sharded: when a shard's
    first job meets its predecessor's final cache entry with another value, every rank reads that off the records,
    the slow protocol runs (re-match, exchange until nothing changes, weights once more) and the result is still the
    unsharded filter's."""
    from synth import Scan
    ctx = pkg.Context(0)
    sc = make_scene(cell_model=2, size=800, scale=0.05, n_beams=360, seed=4)
    ctx.upload_map(1, sc["map"])
    full = sc["scan"]
    reruns = extra_collectives = 0
    for j, pick in enumerate(([10], [200, 201], [100])):
        scan = Scan(full.range[pick], full.angle[pick], np.full(len(pick), 1.0 / len(pick)))
        gp = [0.0, 1e-9, 0.0, 1e-9, 0.0, 0.0, 0.0, 0.0]
        steps = [sc["true_pose"], [0.02, 0.01, 0.0], [0.0, 0.03, 0.01], [0.01, 0.0, 0.0]]
        n, world = 9, 2
        whole = pkg.GmappingFilter(ctx, pkg.gmapping_params(gp8=gp), n, np.arange(2000, 2000 + n, dtype=np.uint32))
        logs, counts = run_loopback_ranks(pkg, world, n, sc, steps, scan, gp, "repair-%d" % j)
        resampled = 0
        for k, d in enumerate(steps):
            rw, iw = whole.step(1, scan.range, scan.angle, None, d, 7 + k)
            resampled += int(rw)
            pw, ww, mw = whole.state()
            np.testing.assert_array_equal(np.concatenate([logs[r][0][k][2] for r in range(world)]), pw)
            np.testing.assert_array_equal(np.concatenate([logs[r][0][k][3] for r in range(world)]), ww)
        reruns += sum(logs[r][0][k][5] for r in range(world) for k in range(len(steps)))
        extra_collectives += logs[0][1]["collectives"] - (1 + len(steps) + resampled)
    assert reruns > 0, "the scene did not exercise a hand-over across the ranks"
    assert extra_collectives > 0  # the repair rounds did run
    ctx.close()


def test_exchange_over_rccl_with_one_rank(pkg):
    """slamhip_shard_exchange on a 1-rank RCCL communicator: ncclSend / ncclRecv to oneself inside one group -- the
    call the map migration makes, on the real transport (with more ranks it needs as many GPUs)."""
    import torch
    ctx = pkg.Context(0)
    ctx.shard_init(0, 1, pkg.shard_unique_id())
    a = torch.arange(4096, dtype=torch.float64, device="cuda") * 1.5
    b = torch.zeros(4096, dtype=torch.float64, device="cuda")
    c = torch.zeros(100, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    ctx.shard_exchange([(0, a.data_ptr(), 4096 * 8), (0, a.data_ptr() + 800, 800)],
                       [(0, b.data_ptr(), 4096 * 8), (0, c.data_ptr(), 800)])
    torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.equal(c, a[100:200])
    assert ctx.shard_p2p_stats() == dict(exchanges=1, bytes_sent=4096 * 8 + 800)
    ctx.shard_exchange([], [])
    ctx.shard_destroy()
    ctx.close()


def run_map_ranks(pkg, world, n, steps_spec, name, fail_rank=None, inject=None):
    """slamhip_gmapping_step_sharded with per-particle maps on `world` in-process ranks; returns per-rank logs:
    (resampled, idx, poses, weights, masters) per step, the final maps of the rank's particles, migration stats"""
    import threading
    import loopback
    from helpers import load
    loopback.lib()
    g = load("gmapping_pf_update.npz")
    w, h = [int(v) for v in g["size"]]
    ox, oy = [int(v) for v in g["origin"]]
    counts = [n // world + (1 if r < n % world else 0) for r in range(world)]
    seeds = np.arange(2000, 2000 + n, dtype=np.uint32)
    logs, errors = [None] * world, []

    def rank_main(rank):
        try:
            ctx = pkg.Context(0, testing=inject is not None)  # (failures are injected through libslamhip_testing.so)
            loopback.attach(pkg, ctx, name, rank, world)
            ctx.map_bind(4, 2, w, h, g["origin"], float(g["scale"]), g["unknown"][:3])
            c0, s0 = pkg.beam_trig(g["step0_angle"])
            ctx.map_append_scan(4, pkg.RULE_GMAPPING, g["step0_delta"], g["step0_range"], c0, s0)
            first = sum(counts[:rank])
            pf = pkg.GmappingFilter(ctx, pkg.gmapping_params(gp8=g["gp"], skip_rate=3, pose_trig=1), n,
                                    seeds[first:first + counts[rank]], first=first, count=counts[rank])
            pf.enable_particle_maps(4, 8, 16 + 24 * n)
            log = []
            dbg = None
            if inject is not None:
                dbg = pkg.load(testing=True).slamhip_gmapping_debug_fail
                dbg.argtypes = [C.c_void_p, C.c_int, C.c_int]
                dbg.restype = C.c_int
            for it, k in enumerate(steps_spec):
                # inject = (rank, where, step): the library's testing hook makes one place fail on one rank as a
                # rank-local error (1 match_finish, 2 the export between the migration's collectives, 3 the final
                # import), armed right before step `step`
                if inject and inject[0] == rank and it == inject[2]:
                    assert dbg(pf.h, inject[1], 1) == 0
                if fail_rank == rank and it == 1:
                    # a step left half done by the caller (match_begin without match_finish): this rank's
                    # step_sharded fails in its own match_begin
                    pf.match_begin(4, g["step%d_range" % k], g["step%d_angle" % k], None, [0.0, 0.0, 0.0])
                try:
                    res, idx = pf.step_sharded(4, g["step%d_range" % k], g["step%d_angle" % k], None,
                                               g["step%d_delta" % k], 7 + it)
                except pkg.SlamHipError as e:
                    log.append(("error", str(e)))
                    continue
                p, wts, m = pf.state()
                log.append((res, np.array(idx).copy(), p, wts, m))
            maps = [pf.particle_map(i, -ox, -oy, w, h) for i in range(counts[rank])]
            logs[rank] = (log, maps, pf.migration_stats(), ctx.shard_p2p_stats())
            pf.close()
            ctx.shard_destroy()
            ctx.close()
        except Exception as e:  # noqa: BLE001
            import traceback
            errors.append((rank, repr(e), traceback.format_exc()))

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert all(not t.is_alive() for t in threads), "a rank is stuck in a collective"
    assert not errors, errors
    return logs, counts, (ox, oy, w, h), g


@pytest.mark.parametrize("world", [2, 3])
def test_step_sharded_with_particle_maps_migrates_inside_the_library(pkg, world):
    """BASELINE configs[4]'s form -- particles WITH their own copy-on-write maps sharded over the ranks -- through the
    library's one entry point: slamhip_gmapping_step_sharded all-gathers weights and records as before and, when a
    resampling draws a particle from another rank, moves that particle's map itself (headers by all-gather, tile
    contents by ONE slamhip_shard_exchange, device to device).  2 and 3 in-process ranks with a tile pool each on
    one GPU against the unsharded filter: poses, weights, masters, resampling indices and EVERY particle's map bit
    for bit through at least two resamplings with migrations (particle_filter.h:83-106,
    lazy_tiled_grid_map.h:40-71)."""
    from helpers import load
    n = 8
    n_base = int(load("gmapping_pf_update.npz")["n_steps"])
    steps_spec = list(range(n_base)) + [1 + (k % (n_base - 1)) for k in range(20)]
    logs, counts, (ox, oy, w, h), g = run_map_ranks(pkg, world, n, steps_spec, "maps-%d" % world)
    ctx = pkg.Context(0)
    ctx.map_bind(4, 2, w, h, g["origin"], float(g["scale"]), g["unknown"][:3])
    c0, s0 = pkg.beam_trig(g["step0_angle"])
    ctx.map_append_scan(4, pkg.RULE_GMAPPING, g["step0_delta"], g["step0_range"], c0, s0)
    whole = pkg.GmappingFilter(ctx, pkg.gmapping_params(gp8=g["gp"], skip_rate=3, pose_trig=1), n,
                               np.arange(2000, 2000 + n, dtype=np.uint32))
    whole.enable_particle_maps(4, 8, 16 + 24 * n)
    resamplings = 0
    for it, k in enumerate(steps_spec):
        res, idx = whole.step(4, g["step%d_range" % k], g["step%d_angle" % k], None, g["step%d_delta" % k], 7 + it)
        pw, ww, mw = whole.state()
        resamplings += int(res)
        for r in range(world):
            assert logs[r][0][it][0] == res, (it, r)
            if res:
                np.testing.assert_array_equal(logs[r][0][it][1], idx)
        np.testing.assert_array_equal(np.concatenate([logs[r][0][it][2] for r in range(world)]), pw, err_msg="step %d" % it)
        np.testing.assert_array_equal(np.concatenate([logs[r][0][it][3] for r in range(world)]), ww)
        np.testing.assert_array_equal(np.concatenate([logs[r][0][it][4] for r in range(world)]), mw)
    first = 0
    for r in range(world):
        for l in range(counts[r]):
            a_p, a_a = whole.particle_map(first + l, -ox, -oy, w, h)
            b_p, b_a = logs[r][1][l]
            np.testing.assert_array_equal(b_p, a_p, err_msg="particle %d" % (first + l))
            np.testing.assert_array_equal(b_a, a_a)
        first += counts[r]
    received = sum(logs[r][2]["maps_received"] for r in range(world))
    sent = sum(logs[r][2]["tile_bytes_sent"] for r in range(world))
    assert resamplings >= 2 and received >= 1, (resamplings, received)
    assert sent == sum(logs[r][3]["bytes_sent"] for r in range(world))
    ctx.close()


def test_a_failing_rank_takes_every_rank_out_of_the_step(pkg):
    """a rank-local error after match_begin used to leave g->pending set (every later step failed with
    'the previous step was not finished') and the other ranks blocked in the all-gather.  Now the failure travels in
    the status word of the step's collective: all ranks return from THAT step with an error, nobody hangs, and the
    filters take the next step."""
    logs, counts, _, _ = run_map_ranks(pkg, 2, 8, [0, 1, 2], "fail-1", fail_rank=1)
    for r in range(2):
        log = logs[r][0]
        assert log[0][0] != "error"
        assert log[1][0] == "error", log[1]
        assert log[2][0] != "error", log[2]
    assert "not finished" in logs[1][0][1][1], logs[1][0][1][1]
    assert "rank 1" in logs[0][0][1][1], logs[0][0][1][1]


def _errors_by_step(logs, world):
    return [[logs[r][0][it][0] == "error" for r in range(world)] for it in range(len(logs[0][0]))]


def test_late_failures_leave_every_collective_together(pkg):
    """ADVICE r3 (medium, twice): a rank that fails BEHIND the step's first collective -- its map update in
    match_finish, the export of a migrating map between the migration's collectives -- used to return while the other
    ranks went on into the resampling's all-gathers and waited there for good.  Now such a rank stays in the step's
    collectives with a status word, and every rank leaves from the same one.  Three ranks in-process (a hang would
    show as a stuck thread), failures injected through the library's testing hook:
      * match_finish in a step that resamples: EVERY rank returns an error from that step, the next step runs;
      * match_finish in a step that does not resample: no collective follows, the rank fails alone and the others
        learn of it from the status word of the NEXT step's first all-gather (everybody fails there, then goes on);
      * the export between the migration's all-gathers: every rank leaves the resampling together;
      * the import behind the migration's last collective: alone, then everybody at the next step."""
    from helpers import load
    world, n = 3, 8
    n_base = int(load("gmapping_pf_update.npz")["n_steps"])
    steps_spec = list(range(n_base)) + [1 + (k % (n_base - 1)) for k in range(20)]
    clean, _, _, _ = run_map_ranks(pkg, world, n, steps_spec, "late-clean")
    res = [bool(clean[0][0][it][0]) for it in range(len(steps_spec))]
    assert any(res[1:]) and not all(res[1:])
    it_res = 1 + res[1:].index(True)
    it_plain = 1 + res[1:-1].index(False)
    # match_finish, resampling step: together, at once
    logs, _, _, _ = run_map_ranks(pkg, world, n, steps_spec, "late-a", inject=(1, 1, it_res))
    err = _errors_by_step(logs, world)
    assert err[it_res] == [True] * world and not any(any(e) for i, e in enumerate(err) if i != it_res), err
    assert "injected" in logs[1][0][it_res][1] and "rank 1" in logs[0][0][it_res][1] and "rank 1" in logs[2][0][it_res][1]
    # match_finish, no resampling: alone, then everybody at the next step's first collective
    logs, _, _, _ = run_map_ranks(pkg, world, n, steps_spec, "late-b", inject=(2, 1, it_plain))
    err = _errors_by_step(logs, world)
    assert err[it_plain] == [False, False, True] and err[it_plain + 1] == [True] * world, err
    assert not any(any(e) for i, e in enumerate(err) if i not in (it_plain, it_plain + 1)), err
    # the migration: export between its collectives (together), import behind them (alone, then everybody)
    logs, _, _, _ = run_map_ranks(pkg, world, n, steps_spec, "late-c", inject=(0, 2, 0))
    err = _errors_by_step(logs, world)
    hit = [i for i, e in enumerate(err) if any(e)]
    assert len(hit) == 1 and err[hit[0]] == [True] * world and res[hit[0]], (hit, err)
    assert "abandoned on every rank" in logs[1][0][hit[0]][1]
    logs, _, _, _ = run_map_ranks(pkg, world, n, steps_spec, "late-d", inject=(1, 3, 0))
    err = _errors_by_step(logs, world)
    hit = [i for i, e in enumerate(err) if any(e)]
    assert len(hit) == 2 and hit[1] == hit[0] + 1 and err[hit[0]] == [False, True, False] and err[hit[1]] == [True] * world, (hit, err)


def test_a_rank_that_stops_responding_takes_the_step_down_within_the_deadline(pkg):
    """the status words cover errors raised BEFORE a collective; a peer that dies INSIDE one used to
    hang the group for good.  Two in-process ranks over the loopback transport (deadline 400 ms): rank 1 takes two
    steps and then never calls again; rank 0's third step must come back with SLAMHIP_ERR_TIMEOUT within the deadline
    -- the library hands the transport's verdict through and abandons the step (the filter is not left `pending`)."""
    import threading
    import time
    import loopback
    from helpers import load
    loopback.lib()
    g = load("gmapping_pf_update.npz")
    w, h = [int(v) for v in g["size"]]
    n, world, name = 8, 2, "dies"
    counts = [4, 4]
    seeds = np.arange(2000, 2000 + n, dtype=np.uint32)
    out, errors = {}, []
    attached = threading.Barrier(world)

    def rank_main(rank):
        try:
            ctx = pkg.Context(0)
            loopback.attach(pkg, ctx, name, rank, world)
            attached.wait()
            if rank == 0:
                loopback.set_timeout(name, 400)
            ctx.map_bind(4, 2, w, h, g["origin"], float(g["scale"]), g["unknown"][:3])
            c0, s0 = pkg.beam_trig(g["step0_angle"])
            ctx.map_append_scan(4, pkg.RULE_GMAPPING, g["step0_delta"], g["step0_range"], c0, s0)
            first = sum(counts[:rank])
            pf = pkg.GmappingFilter(ctx, pkg.gmapping_params(gp8=g["gp"], skip_rate=3, pose_trig=1), n,
                                    seeds[first:first + counts[rank]], first=first, count=counts[rank])
            for it in range(2 if rank == 1 else 3):
                t0 = time.time()
                try:
                    pf.step_sharded(4, g["step%d_range" % it], g["step%d_angle" % it], None, g["step%d_delta" % it], 7 + it)
                    out[(rank, it)] = ("ok", time.time() - t0)
                except pkg.SlamHipError as e:
                    out[(rank, it)] = (str(e), time.time() - t0)
            if rank == 0:
                # the step was abandoned, not left half done: a host-side call that needs no group still works
                pf.match_abort()
                out["state"] = pf.state()[0].shape
            pf.close()
            ctx.shard_destroy()
            ctx.close()
        except Exception as e:  # noqa: BLE001
            import traceback
            errors.append((rank, repr(e), traceback.format_exc()))

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert all(not t.is_alive() for t in threads), "a rank is stuck in a collective"
    assert not errors, errors
    assert out[(0, 0)][0] == out[(0, 1)][0] == out[(1, 0)][0] == out[(1, 1)][0] == "ok"
    msg, took = out[(0, 2)]
    assert msg != "ok" and took < 3.0, out[(0, 2)]
    assert out["state"] == (4, 3)


def test_rccl_collective_wait_is_bounded(pkg):
    """The same on the built-in transport: slamhip_shard_allgather / _exchange poll the stream until the group's
    deadline instead of hipStreamSynchronize.  One GPU admits one RCCL rank, so the dead peer is played by a kernel
    that keeps the stream busy for a second (testing hook) in front of the collective of a 1-rank group with a 150 ms
    deadline: SLAMHIP_ERR_TIMEOUT within the deadline, the communicator aborted, every later collective refused at
    once -- and after slamhip_shard_destroy a fresh group works."""
    import time
    ctx = pkg.Context(0, testing=True)
    L = pkg.load(testing=True)
    L.slamhip_debug_stall.argtypes = [C.c_void_p, C.c_int]
    L.slamhip_debug_stall.restype = C.c_int
    ctx.shard_init(0, 1, pkg.shard_unique_id())
    a = np.arange(12, dtype=np.float64).reshape(4, 3)
    np.testing.assert_array_equal(ctx.shard_allgather(a, [4]), a)
    ctx.shard_set_timeout(150)
    assert L.slamhip_debug_stall(ctx.h, 1000) == 0
    t0 = time.time()
    with pytest.raises(pkg.SlamHipError, match="stopped responding"):
        ctx.shard_allgather(a, [4])
    assert time.time() - t0 < 0.8
    with pytest.raises(pkg.SlamHipError, match="broken"):
        ctx.shard_allgather(a, [4])
    with pytest.raises(pkg.SlamHipError, match="broken"):
        ctx.shard_exchange([], [])
    ctx.shard_destroy()
    ctx.synchronize()
    ctx.shard_init(0, 1, pkg.shard_unique_id())
    np.testing.assert_array_equal(ctx.shard_allgather(a, [4]), a)
    ctx.shard_destroy()
    ctx.close()


def _rank_main_maps(rank, world, uid_path, out_path):
    """one process = one GPU = one rank of an RCCL group: the sharded step with per-particle maps"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import time
    import __graft_entry__ as g2
    from helpers import load
    pkg = g2.load_package()
    if rank == 0:
        np.save(uid_path + ".tmp.npy", pkg.shard_unique_id())
        os.replace(uid_path + ".tmp.npy", uid_path)
    else:
        for _ in range(600):
            if os.path.exists(uid_path):
                break
            time.sleep(0.05)
    uid = np.load(uid_path)
    g = load("gmapping_pf_update.npz")
    w, h = [int(v) for v in g["size"]]
    ox, oy = [int(v) for v in g["origin"]]
    n = 8
    counts = [n // world + (1 if r < n % world else 0) for r in range(world)]
    first = sum(counts[:rank])
    ctx = pkg.Context(rank)
    ctx.shard_init(rank, world, uid)
    ctx.map_bind(4, 2, w, h, g["origin"], float(g["scale"]), g["unknown"][:3])
    c0, s0 = pkg.beam_trig(g["step0_angle"])
    ctx.map_append_scan(4, pkg.RULE_GMAPPING, g["step0_delta"], g["step0_range"], c0, s0)
    seeds = np.arange(2000, 2000 + n, dtype=np.uint32)
    pf = pkg.GmappingFilter(ctx, pkg.gmapping_params(gp8=g["gp"], skip_rate=3, pose_trig=1), n,
                            seeds[first:first + counts[rank]], first=first, count=counts[rank])
    pf.enable_particle_maps(4, 8, 16 + 24 * n)
    n_base = int(g["n_steps"])
    log = []
    for it, k in enumerate(list(range(n_base)) + [1 + (j % (n_base - 1)) for j in range(20)]):
        res, idx = pf.step_sharded(4, g["step%d_range" % k], g["step%d_angle" % k], None, g["step%d_delta" % k], 7 + it)
        p, wts, m = pf.state()
        log.append((res, np.array(idx).copy(), p, wts, m))
    maps = [pf.particle_map(i, -ox, -oy, w, h) for i in range(counts[rank])]
    np.save(out_path % rank, np.array([log, maps, pf.migration_stats()], dtype=object), allow_pickle=True)
    pf.close()
    ctx.shard_destroy()
    ctx.close()


def test_two_ranks_over_rccl_with_particle_maps(pkg, tmp_path):
    """The map migration on its real transport: two processes, two GPUs, ncclSend / ncclRecv of the tile contents
    between them -- against the unsharded filter on one GPU, every particle's map bit for bit.  Needs two GPUs."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    import multiprocessing as mp
    from helpers import load
    mpc = mp.get_context("spawn")
    uid_path, out_path = str(tmp_path / "uid.npy"), str(tmp_path / "rank%d.npy")
    procs = [mpc.Process(target=_rank_main_maps, args=(r, 2, uid_path, out_path)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    got = [np.load(out_path % r, allow_pickle=True) for r in range(2)]
    g = load("gmapping_pf_update.npz")
    w, h = [int(v) for v in g["size"]]
    ox, oy = [int(v) for v in g["origin"]]
    n = 8
    ctx = pkg.Context(0)
    ctx.map_bind(4, 2, w, h, g["origin"], float(g["scale"]), g["unknown"][:3])
    c0, s0 = pkg.beam_trig(g["step0_angle"])
    ctx.map_append_scan(4, pkg.RULE_GMAPPING, g["step0_delta"], g["step0_range"], c0, s0)
    whole = pkg.GmappingFilter(ctx, pkg.gmapping_params(gp8=g["gp"], skip_rate=3, pose_trig=1), n,
                               np.arange(2000, 2000 + n, dtype=np.uint32))
    whole.enable_particle_maps(4, 8, 16 + 24 * n)
    n_base = int(g["n_steps"])
    for it, k in enumerate(list(range(n_base)) + [1 + (j % (n_base - 1)) for j in range(20)]):
        res, idx = whole.step(4, g["step%d_range" % k], g["step%d_angle" % k], None, g["step%d_delta" % k], 7 + it)
        pw, ww, mw = whole.state()
        assert got[0][0][it][0] == got[1][0][it][0] == res
        np.testing.assert_array_equal(np.concatenate([got[r][0][it][2] for r in range(2)]), pw)
        np.testing.assert_array_equal(np.concatenate([got[r][0][it][3] for r in range(2)]), ww)
    for i in range(n):
        a_p, a_a = whole.particle_map(i, -ox, -oy, w, h)
        b_p, b_a = got[i // 4][1][i % 4]
        np.testing.assert_array_equal(b_p, a_p)
        np.testing.assert_array_equal(b_a, a_a)
    assert got[0][2]["maps_received"] + got[1][2]["maps_received"] >= 1
    ctx.close()
