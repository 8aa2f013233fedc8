"""bench_legs.particle_filter -- BASELINE configs[3] and [4]: the GMapping filter legs (likelihood step sharded over the
ranks, the shared-map step, per-particle maps on one GPU and sharded, cfg5 on one GPU and sharded)."""
import time

import numpy as np

from .common import BYTES_PER_UNIT, HBM_PEAK_GBS, k6_roofline, load_traffic, roofline_valu


def resampling_deltas(sc, n_steps, seed=9):
    """Odometry increments that take the filter THROUGH a resampling within ten steps (measured on MI355X, 13 ... 100
    particles: the first one in step 7 or 8): large steps that do NOT cancel -- the travelled distance leaves
    try_resample's `sq_dist <= 0.5 && theta <= 0.2` region (gmapping_particle_filter.h:88-99), and with the robot
    reported a metre off the ground the scan was taken from the scan probabilities spread the weights until
    2 / sum(w^2) < N (particle_filter.h:34-43); tests/test_gpu_shard.py drives its filters the same way.  Steps that
    return the robot to where it was never resample: the hill climbing pulls every particle back and the weights stay
    level (measured: 0 resamplings in 13 steps).  First entry: the pose the filter starts from."""
    cyc = [[0.02, 0.01, 0.01], [0.4, 0.5, 0.3], [0.01, -0.02, 0.02], [0.5, -0.4, 0.25], [0.02, 0.02, 0.0], [0.45, 0.5, -0.3],
           [0.0, 0.01, 0.01]]
    return [sc["true_pose"]] + [np.array(cyc[k % len(cyc)]) for k in range(n_steps)]


def sharded_particle_maps_leg(args, pkg, ctx, gather, counts, gp, seeds, n, first, count, rank, world, scan, deltas,
                              dist, torch, dev, map_id=1, size=None, tiles_per_particle=None, adder=None, steps=None):
    """Per-particle copy-on-write maps with the particles sharded over the ranks.  With the context in the library's
    RCCL group (--backend nccl) one call per scan does everything: slamhip_gmapping_step_sharded matches the shard,
    all-gathers carry records + weights, plans the resampling identically everywhere and, when a resampling draws a
    particle from another rank, moves its map itself (headers by all-gather, tile contents by ONE ncclSend/ncclRecv
    group, device to device).  Under --backend gloo (ranks sharing GPUs: path testing) the same protocol runs over
    torch.distributed: all-gather of the raw weights, records on resampling, maps point to point through host
    buffers."""
    size = size or args.pf_size
    tiles_per_particle = tiles_per_particle or args.pf_tiles_per_particle
    pfm = pkg.GmappingFilter(ctx, pkg.gmapping_params(gp8=gp), n, seeds, first=first, count=count)
    ext = (size + 127) // 128 + 1
    # room for the shard's own maps and as much again for maps that migrate in; never more than 70 % of what the GPU
    # has free (ranks that share a GPU -- path testing under gloo -- would otherwise take each other's memory)
    pool_tiles = ext * ext + 2 * count * tiles_per_particle
    free_bytes, _total = torch.cuda.mem_get_info()
    sharing = max(1, -(-world // max(1, torch.cuda.device_count())))  # ranks on this GPU (they size their pools at once)
    pool_tiles = max(ext * ext + count, min(pool_tiles, int(0.7 * free_bytes / sharing / (128 * 128 * 48))))
    pfm.enable_particle_maps(map_id, extent_tiles=ext, pool_tiles=pool_tiles, **(adder or {}))
    bounds = np.cumsum(counts)
    owner = lambda j: int(np.searchsorted(bounds, int(j), side="right"))  # noqa: E731  (contiguous blocks)
    moved_bytes = 0
    resamplings = 0
    in_library = args.backend == "nccl"

    def one(k):
        nonlocal moved_bytes, resamplings
        if in_library:
            req, _ = pfm.step_sharded(map_id, scan.range, scan.angle, None, deltas[k % len(deltas)], 7 + k)
            resamplings += 1 if req else 0
            return
        raw = pfm.predict_match(map_id, scan.range, scan.angle, None, deltas[k % len(deltas)])
        req, idx = pfm.plan_resample(gather(raw, torch.float64), 7 + k)
        if not req:
            return
        resamplings += 1
        blobs = gather(pfm.export(), torch.uint8)
        pairs = sorted({(int(idx[i]), owner(i)) for i in range(n) if owner(idx[i]) != owner(i)})
        mine = {j: pfm.export_particle_map(j - first) for j in sorted({j for j, _ in pairs if owner(j) == rank})}
        sizes = np.zeros(n, np.int64)
        for j, b in mine.items():
            sizes[j] = b.size
        sizes = gather(sizes[first:first + count], torch.int64)
        ops, recv = [], {}
        for j, r in pairs:
            if owner(j) == rank:
                t = torch.from_numpy(mine[j]).to(dev)
                ops.append(dist.P2POp(dist.isend, t, r))
                moved_bytes += int(t.numel())
            elif r == rank:
                recv[j] = torch.empty(int(sizes[j]), dtype=torch.uint8, device=dev)
                ops.append(dist.P2POp(dist.irecv, recv[j], owner(j)))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        pfm.import_maps(blobs, idx, {j: t.cpu().numpy() for j, t in recv.items()})

    one(0)
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    msteps = steps or max(3, args.pf_steps)
    for k in range(1, 1 + msteps):
        one(k)
    dist.barrier()
    torch.cuda.synchronize()
    dm = time.perf_counter() - t0
    maps_received = 0
    p2p = None
    if in_library:
        mg = pfm.migration_stats()
        moved_bytes, maps_received = mg["tile_bytes_sent"], mg["maps_received"]
        p2p = ctx.shard_p2p_stats()
    tt = torch.tensor([dm, float(moved_bytes)], dtype=torch.float64, device=dev)
    mx = tt.clone()
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    sm = tt.clone()
    dist.all_reduce(sm, op=dist.ReduceOp.SUM)
    st = pfm.particle_map_stats()
    out = {"value": n * msteps / mx[0].item(), "unit": "particles/s", "ms_per_step": 1e3 * mx[0].item() / msteps,
           "steps": msteps, "resamplings": resamplings, "map_bytes_moved_between_ranks": sm[1].item(),
           "maps_received_rank0": maps_received, "shard_p2p_stats_rank0": p2p,
           "tiles_in_use_rank0": st["tiles_in_use"], "ranks": world, "scaling": "strong",
           "migration": ("inside slamhip_gmapping_step_sharded: headers all-gathered, tile contents in one RCCL "
                         "send/recv group, device to device" if in_library else
                         "torch.distributed over gloo: batch_isend_irecv of exported host buffers"),
           "note": "particles and their copy-on-write maps sharded over %d ranks; maps migrate point to point "
                   "on resampling" % world}
    pfm.close()
    return out


def cfg5_sharded_leg(args, pkg, ctx, rank, world, dist, torch):
    """BASELINE configs[4] in the form BASELINE states it: `--cfg5-particles` particles WITH their own maps sharded over
    the ranks of one node (8000x8000 @ 0.025 m, area occupancy estimator, blur 0.1 m, map update fused behind the
    likelihood), every rank its own tile pool, maps migrating over xGMI on resampling -- through the library's one
    entry point per scan."""
    from synth import make_scene
    n, size, scale = args.cfg5_particles, args.cfg5_size, args.cfg5_scale
    if n < world:
        return {"skipped": "fewer particles (%d) than ranks (%d)" % (n, world)}
    win = min(size, 3200)
    sc = make_scene(cell_model=2, size=win, scale=scale, n_beams=args.beams, seed=6, blur_m=0.1)
    m, scan = sc["map"], sc["scan"]
    off = (size - win) // 2
    ctx.map_bind(2, 2, size, size, (m.origin[0] + off, m.origin[1] + off), scale, m.unknown)
    ctx.map_upload_window(2, off, off, m.payload)
    counts = [n // world + (1 if r < n % world else 0) for r in range(world)]
    first, count = sum(counts[:rank]), counts[rank]
    seeds = np.arange(3000, 3000 + n, dtype=np.uint32)[first:first + count]
    gp = [0.0, args.pf_sigma_xy / 2, 0.0, args.pf_sigma_th, 0.0, 0.0, 0.0, 0.0]
    deltas = resampling_deltas(sc, args.cfg5_steps + 4, seed=8)  # (through resamplings: maps migrate between the ranks)
    reach = int(np.ceil(2.0 * (float(scan.range.max()) + 1.0) / scale / 128.0)) + 2
    dev = args.coll_device

    def gather(a, dtype):
        a = np.ascontiguousarray(a)
        per = a.size // count
        padded = np.zeros(max(counts) * per, dtype=a.dtype)
        padded[:a.size] = a.ravel()
        t = torch.from_numpy(padded).to(dev)
        out = torch.empty(world * t.numel(), dtype=dtype, device=dev)
        dist.all_gather_into_tensor(out, t)
        out = out.cpu().numpy().reshape(world, -1)
        return np.concatenate([out[r, :counts[r] * per] for r in range(world)])

    out = sharded_particle_maps_leg(args, pkg, ctx, gather, counts, gp, seeds, n, first, count, rank, world, scan, deltas,
                                    dist, torch, dev, map_id=2, size=size, tiles_per_particle=reach * reach,
                                    adder=dict(blur=0.1, estimator=1, shift_amount=0.01 * scale), steps=args.cfg5_steps)
    out["metric"] = "particles/sec at N=%d" % n
    out["workload"] = ("cfg5: GMapping %d particles sharded over %d GPUs, %d beams, %dx%d @%.3f m, per-particle "
                       "copy-on-write maps (a tile pool per rank), area occupancy estimator + blur 0.1 m map update in one "
                       "batched K6 per rank and step" % (n, world, scan.n, size, size, scale))
    ctx.map_release(2)
    return out


def join_shard_group(args, pkg, ctx, rank, world, dist, torch):
    """The context joins the library's RCCL group (once): torch.distributed only carries the 128-byte id."""
    if getattr(args, "_joined", False):
        return
    dev = args.coll_device
    uid = torch.from_numpy(pkg.shard_unique_id() if rank == 0 else np.zeros(pkg.SHARD_ID_BYTES, np.uint8)).to(dev)
    dist.broadcast(uid, 0)
    ctx.shard_init(rank, world, uid.cpu().numpy())
    args._joined = True


def particle_filter_leg(args, pkg, ctx, sc, rank, world, dist, torch):
    """BASELINE cfg 4: GMapping filter, `--particles` particles sharded over the ranks (contiguous
    blocks), 1080-beam scan, 4000x4000 @0.05 m GMapping-cell map replicated per GPU (the reference's
    particles share one map), HC(6, 0.1, 0.1), gate open so every particle matches on every scan.
    One collective per step: all-gather of the raw weights over RCCL, inside the library (plus the particle
    records when a resampling happens).  Strong scaling: the particle count is fixed."""
    n = args.particles
    if n < world:
        return {"skipped": "fewer particles (%d) than ranks (%d)" % (n, world)}
    legs = args.leg_set
    ctx.upload_map(1, sc["map"])
    # contiguous blocks; the first n % world ranks hold one particle more (100 particles on 8 GPUs: 13 x 4 + 12 x 4)
    counts = [n // world + (1 if r < n % world else 0) for r in range(world)]
    firsts = [sum(counts[:r]) for r in range(world)]
    count, first = counts[rank], firsts[rank]
    seeds = np.arange(1000, 1000 + n, dtype=np.uint32)[first:first + count]
    gp = [0.0, args.pf_sigma_xy, 0.0, args.pf_sigma_th, 0.0, 0.0, 0.0, 0.0]
    scan = sc["scan"]
    dev = args.coll_device

    def gather(a, dtype):
        """all-gather of per-particle rows over torch.distributed (gloo path; uneven shards are padded)"""
        if world == 1:
            return np.asarray(a)
        a = np.ascontiguousarray(a)
        per = a.size // count
        padded = np.zeros(max(counts) * per, dtype=a.dtype)
        padded[:a.size] = a.ravel()
        t = torch.from_numpy(padded).to(dev)
        out = torch.empty(world * t.numel(), dtype=dtype, device=dev)
        dist.all_gather_into_tensor(out, t)
        out = out.cpu().numpy().reshape(world, -1)
        return np.concatenate([out[r, :counts[r] * per] for r in range(world)])

    rs = np.random.RandomState(5)
    deltas = [sc["true_pose"]] + [rs.randn(3) * [0.05, 0.05, 0.02] for _ in range(max(args.pf_steps, 10) + 6)]
    out = {"metric": "particles/sec at N=%d" % n, "unit": "particles/s", "scaling": "strong", "ranks": world}
    # the data-path collective lives in the library (csrc/shard.cpp: RCCL group per context, all-gather of
    # the raw weights inside slamhip_gmapping_step_sharded); torch.distributed only carries the 128-byte
    # group id to the ranks and the benchmark's own barrier / max-over-ranks
    in_library = world > 1 and args.backend == "nccl"
    if in_library:
        join_shard_group(args, pkg, ctx, rank, world, dist, torch)

    if "pf" in legs:
        pf = pkg.GmappingFilter(ctx, pkg.gmapping_params(gp8=gp), n, seeds, first=first, count=count)
        calls = 0
        resamplings = 0

        def one(k):
            nonlocal calls, resamplings
            if in_library:
                req, _idx = pf.step_sharded(1, scan.range, scan.angle, None, deltas[k], 7 + k)
                calls += pf.stats()["scorer_calls"]
                resamplings += 1 if req else 0
                return
            raw = pf.predict_match(1, scan.range, scan.angle, None, deltas[k])
            calls += pf.stats()["scorer_calls"]
            allw = gather(raw, torch.float64)
            req, idx = pf.plan_resample(allw, 7 + k)
            if req:
                resamplings += 1
                pf.import_(gather(pf.export(), torch.uint8), idx)

        for k in range(2):
            one(k)
        calls = 0
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(2, 2 + args.pf_steps):
            one(k)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = pf.stats()
        # second, instrumented pass (HIP events attached to every K3 dispatch; never in the timed pass)
        ctx.profile_enable(True)
        ctx.profile_read(reset=True)
        t1 = time.perf_counter()
        for k in range(2 + args.pf_steps, 2 + args.pf_steps + 3):
            one(k)
        ctx.synchronize()
        dt_instr = time.perf_counter() - t1
        ctx.profile_enable(False)
        g_ms, g_launches, g_units = ctx.profile_read(reset=True)
        bpu = BYTES_PER_UNIT["gmapping"]
        g_achieved = g_units * bpu / (g_ms * 1e-3) / 1e9 if g_ms > 0 else 0.0
        pf_traffic, pf_traffic_src = load_traffic("pf")
        pf_roofline = {"bound": "hbm", "achieved": g_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": g_achieved / HBM_PEAK_GBS, "traffic": pf_traffic, "traffic_source": pf_traffic_src,
                       "kernel": "k_hc_chain_resident_gm" if g_launches <= 3 * 2 else "k_hc_chain_step",
                       "kernel_note": "one hill-climbing chain per particle on the device (GMapping OOPE: K3's one-pose "
                                      "body): ONE co-resident launch per step when all chains' workgroups fit the device "
                                      "(csrc/hc_resident_gm.hip), else shared launches per super-step (csrc/hc_chain.hip)",
                       "bytes_note": "algorithmic bytes = SURVEY 8d's 232 per (pose, beam): the beam record and the nine "
                                     "32-byte cells of the GMapping window.  Since r05 the kernel reads ONE 4-byte "
                                     "neighbourhood mask per beam and the obstacle means of the full cells only (DESIGN 3, "
                                     "K3): `traffic` is what it really moves",
                       "bytes_per_unit": bpu, "launches": g_launches, "units_launched": g_units,
                       "avg_launch_us": 1e3 * g_ms / max(g_launches, 1),
                       "kernel_busy_frac": g_ms / (1e3 * dt_instr) if dt_instr > 0 else None,
                       "timing": "HIP events attached to each dispatch, 3 extra steps after the timed pass"}
        if world > 1:
            tt = torch.tensor([dt, float(calls)], dtype=torch.float64, device=dev)
            mx = tt.clone()
            dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            sm = tt.clone()
            dist.all_reduce(sm, op=dist.ReduceOp.SUM)
            dt, calls = mx[0].item(), sm[1].item()
        collective = "none (1 rank)"
        if in_library:
            ss = ctx.shard_stats()
            collective = ("slamhip_shard_allgather inside slamhip_gmapping_step_sharded: RCCL through the C-ABI, %d "
                          "ranks in the group, %d collectives / %d bytes on this rank over the run"
                          % (ctx.shard_info()[1], ss["collectives"], ss["bytes"]))
        elif world > 1:
            collective = "all_gather(raw weights) per step over gloo (torch.distributed; ranks share GPUs)"
        out.update(value=n * args.pf_steps / dt, ms_per_step=1e3 * dt / args.pf_steps, steps=args.pf_steps,
                   roofline=pf_roofline, roofline_valu=roofline_valu("pf", pf_roofline["avg_launch_us"]),
                   pose_candidates_beams_per_s=calls * scan.n / dt,
                   workload="cfg4: GMapping %d particles sharded over %d GPU(s), %d beams, %dx%d @%.2f m "
                            "GMapping cell, HC(6,0.1,0.1), likelihood step without map update"
                            % (n, world, scan.n, args.pf_size, args.pf_size, args.scale),
                   collective=collective, launches_last_step=st["launches"],
                   carry_reruns_last_step=st["carry_reruns"], resamplings=resamplings)
        pf.close()
        if world == 1:
            # What the 1/2/4/8-GPU strong-scaling curve should look like, stated before it is measured (the driver
            # runs it): the step time of the LARGEST shard of a G-rank run -- ceil(n / G)
            # particles of the n, measured on this GPU -- plus the step's one collective (measured on a 1-rank RCCL
            # group here: host -> device -> ncclAllGather -> device -> host; more ranks add link latency to it).
            try:
                model = []
                coll_us = None
                try:
                    ctx.shard_init(0, 1, pkg.shard_unique_id())
                    blk = np.zeros((n, 1))
                    for _ in range(5):
                        ctx.shard_allgather(blk, [n])
                    tc = time.perf_counter()
                    for _ in range(50):
                        ctx.shard_allgather(blk, [n])
                    coll_us = 1e6 * (time.perf_counter() - tc) / 50
                    ctx.shard_destroy()
                except Exception as e:  # noqa: BLE001
                    coll_us = None
                    model.append({"collective_error": str(e)})
                for G in (1, 2, 4, 8):
                    if n < G:
                        continue
                    cG = -(-n // G)
                    f = pkg.GmappingFilter(ctx, pkg.gmapping_params(gp8=gp), n,
                                           np.arange(1000, 1000 + n, dtype=np.uint32)[:cG], first=0, count=cG)
                    for k in range(2):
                        f.predict_match(1, scan.range, scan.angle, None, deltas[k])
                    ctx.synchronize()
                    tg = time.perf_counter()
                    for k in range(2, 2 + args.pf_steps):
                        f.predict_match(1, scan.range, scan.angle, None, deltas[k])
                    ctx.synchronize()
                    shard_ms = 1e3 * (time.perf_counter() - tg) / args.pf_steps
                    f.close()
                    pred = shard_ms + (coll_us or 0.0) * 1e-3 * (1 if G > 1 else 0)
                    model.append({"ranks": G, "particles_on_largest_shard": cG, "shard_ms_per_step": shard_ms,
                                  "predicted_ms_per_step": pred, "predicted_particles_per_s": n / (pred * 1e-3),
                                  "predicted_speedup": None})
                base = next((m_["predicted_ms_per_step"] for m_ in model if m_.get("ranks") == 1), None)
                for m_ in model:
                    if base and "ranks" in m_:
                        m_["predicted_speedup"] = base / m_["predicted_ms_per_step"]
                out["scaling_model"] = {
                    "by_ranks": model, "collective_us_one_rank_group": coll_us,
                    "note": "strong scaling of a latency chain: a shard's step costs about as many super-steps as the "
                            "whole filter's (every particle's accept chain is as long), only narrower launches -- so "
                            "the curve flattens early; measured per-shard times on one GPU + the step's one all-gather"}
            except pkg.SlamHipError as e:
                out["scaling_model"] = {"error": str(e)}
    if world == 1 and "pf_update" in legs:
        # the reference's full step: each particle appends its scan to the shared map before the
        # next one matches (sequential by construction) -- a few steps are enough
        pfu = pkg.GmappingFilter(ctx, pkg.gmapping_params(gp8=gp), n, seeds)
        pfu.set_map_update(True)
        pfu.step(1, scan.range, scan.angle, None, deltas[0], 7)
        torch.cuda.synchronize()
        tu = time.perf_counter()
        ksteps = max(10, args.pf_steps)  # (r05 timed 3: VERDICT r5 "What's weak" 5)
        for k in range(1, 1 + ksteps):
            pfu.step(1, scan.range, scan.angle, None, deltas[k], 7 + k)
        torch.cuda.synchronize()
        du = time.perf_counter() - tu
        ctx.profile_enable(True)
        ctx.profile_read(reset=True)
        ctx.profile_read_map_update(reset=True)
        pfu.step(1, scan.range, scan.angle, None, deltas[ksteps + 1], 7 + ksteps + 1)
        ctx.synchronize()
        ctx.profile_enable(False)
        ctx.profile_read(reset=True)
        out["with_map_update"] = {"value": n * ksteps / du, "unit": "particles/s", "ms_per_step": 1e3 * du / ksteps,
                                  "steps": ksteps,
                                  "note": "sequential particles: GPU match then K6 map update on the shared map, as "
                                          "the reference does",
                                  "roofline_map_update": k6_roofline(ctx, "one extra step after the timed pass (%d "
                                                                          "single-scan updates)" % n, "pf_update")}
        pfu.close()
    if world == 1 and "pf_maps" in legs:
        # per-particle copy-on-write maps (tile pool): lock-step matching on every
        # particle's own map + ONE batched K6 for all appends of the step
        try:
            pfm = pkg.GmappingFilter(ctx, pkg.gmapping_params(gp8=gp), n, seeds)
            ext = (args.pf_size + 127) // 128 + 1
            pfm.enable_particle_maps(1, extent_tiles=ext, pool_tiles=ext * ext + n * args.pf_tiles_per_particle)
            pfm.step(1, scan.range, scan.angle, None, deltas[0], 7)  # first step clones every touched tile
            first_stats = pfm.particle_map_stats()
            torch.cuda.synchronize()
            tm = time.perf_counter()
            msteps = max(3, args.pf_steps)
            for k in range(1, 1 + msteps):
                pfm.step(1, scan.range, scan.angle, None, deltas[k % len(deltas)], 7 + k)
            torch.cuda.synchronize()
            dm = time.perf_counter() - tm
            stt = pfm.particle_map_stats()
            ctx.profile_enable(True)
            ctx.profile_read(reset=True)
            ctx.profile_read_map_update(reset=True)
            for k in range(2):
                pfm.step(1, scan.range, scan.angle, None, deltas[(msteps + 1 + k) % len(deltas)], 7 + msteps + 1 + k)
            ctx.synchronize()
            ctx.profile_enable(False)
            ctx.profile_read(reset=True)
            out["with_particle_maps"] = {
                "value": n * msteps / dm, "unit": "particles/s", "ms_per_step": 1e3 * dm / msteps,
                "steps": msteps, "tiles_in_use": stt["tiles_in_use"], "pool_bytes": stt["bytes"],
                "cow_copies_first_step": first_stats["cow_copies"], "cow_copies_total": stt["cow_copies"],
                "cell_updates_last_step": stt["cell_updates"],
                "note": "every particle owns a copy-on-write map (128x128-cell tiles); matching in "
                        "lock-step, map updates of all particles in one batched K6",
                "roofline_map_update": k6_roofline(ctx, "2 extra steps after the timed pass", "pf_maps")}
            pfm.close()
            # ... and what the sharded form of this leg (the default at --gpus N > 1) should cost: the step of a filter
            # of ceil(n / G) particles with their own maps, through the SAME resampling-crossing odometry the sharded leg
            # is driven with -- a shard's matching, its batched K6 and its resampling copies; the cross-rank migrations
            # (tile bodies over xGMI) are NOT in it
            try:
                msteps2 = max(10, args.pf_steps)
                dl = resampling_deltas(sc, msteps2 + 2)
                model = []
                for G in (1, 2, 4, 8):
                    if n < G:
                        continue
                    cG = -(-n // G)
                    f = pkg.GmappingFilter(ctx, pkg.gmapping_params(gp8=gp), cG, np.arange(1000, 1000 + cG, dtype=np.uint32))
                    f.enable_particle_maps(1, extent_tiles=ext, pool_tiles=ext * ext + 2 * cG * args.pf_tiles_per_particle)
                    f.step(1, scan.range, scan.angle, None, dl[0], 7)
                    torch.cuda.synchronize()
                    tg = time.perf_counter()
                    rsm = 0
                    for k in range(1, 1 + msteps2):
                        rq, _ = f.step(1, scan.range, scan.angle, None, dl[k], 7 + k)
                        rsm += 1 if rq else 0
                    torch.cuda.synchronize()
                    shard_ms = 1e3 * (time.perf_counter() - tg) / msteps2
                    f.close()
                    model.append({"ranks": G, "particles_on_largest_shard": cG, "shard_ms_per_step": shard_ms,
                                  "resamplings": rsm, "predicted_particles_per_s": n / (shard_ms * 1e-3)})
                base = model[0]["shard_ms_per_step"] if model else None
                for m_ in model:
                    m_["predicted_speedup"] = base / m_["shard_ms_per_step"]
                out["with_particle_maps"]["scaling_model"] = {
                    "by_ranks": model, "steps": msteps2,
                    "note": "per-particle maps, odometry that takes the filter through a resampling: the batched map "
                            "update and the tile traffic shrink with the shard, so this leg scales where the likelihood-only "
                            "chains do not; + one all-gather per step and the migrating maps' tiles over xGMI (not modelled)"}
            except pkg.SlamHipError as e:
                out["with_particle_maps"]["scaling_model"] = {"error": str(e)}
        except pkg.SlamHipError as e:  # e.g. the pool does not fit: report, do not hide
            out["with_particle_maps"] = {"error": str(e)}
    if world > 1 and args.pf_maps_sharded:
        # what "100 particles sharded across GPUs" means once maps are updated: every particle its
        # own copy-on-write map, one batched K6 per rank and step, maps migrating on resampling -- driven through
        # resamplings on purpose (the likelihood leg above never leaves the gate)
        msteps = max(10, args.pf_steps)
        out["with_particle_maps"] = sharded_particle_maps_leg(args, pkg, ctx, gather, counts, gp, seeds, n, first, count,
                                                              rank, world, scan, resampling_deltas(sc, msteps + 2), dist,
                                                              torch, dev, steps=msteps)
    ctx.map_release(1)
    return out


def cfg5_leg(args, pkg, ctx, torch):
    """BASELINE configs[4] on ONE GPU: `--cfg5-particles` particles, 8000x8000 @ 0.025 m GMapping-cell map,
    AreaOccupancyEstimator + blur 0.1 m ray-trace update, every particle its own copy-on-write map: lock-step
    likelihood (K3 through tile tables) + one batched K6 per step.  The 8000^2 dense ancestor is bound in HBM
    (2 GB + 1 GB of counters) and only the window the synthetic world covers is uploaded."""
    from synth import make_scene
    n, size, scale = args.cfg5_particles, args.cfg5_size, args.cfg5_scale
    win = min(size, 3200)  # 80 m of world at 0.025 m: the rooms + corridors raster is at most ~56 m across
    t0 = time.perf_counter()
    sc = make_scene(cell_model=2, size=win, scale=scale, n_beams=args.beams, seed=6, blur_m=0.1)
    t_scene = time.perf_counter() - t0
    m, scan = sc["map"], sc["scan"]
    off = (size - win) // 2
    ctx.map_bind(2, 2, size, size, (m.origin[0] + off, m.origin[1] + off), scale, m.unknown)
    ctx.map_upload_window(2, off, off, m.payload)
    gp = [0.0, args.pf_sigma_xy / 2, 0.0, args.pf_sigma_th, 0.0, 0.0, 0.0, 0.0]
    pf = pkg.GmappingFilter(ctx, pkg.gmapping_params(gp8=gp), n, np.arange(3000, 3000 + n, dtype=np.uint32))
    ext = (size + 127) // 128 + 1
    reach = int(np.ceil(2.0 * (float(scan.range.max()) + 1.0) / scale / 128.0)) + 2
    per_particle = reach * reach
    try:
        pf.enable_particle_maps(2, extent_tiles=ext, pool_tiles=ext * ext + n * per_particle, blur=0.1, estimator=1,
                                shift_amount=0.01 * scale)
    except pkg.SlamHipError as e:
        pf.close()
        ctx.map_release(2)
        return {"error": "tile pool: %s" % e}
    rs = np.random.RandomState(8)
    deltas = [sc["true_pose"]] + [rs.randn(3) * [0.03, 0.03, 0.01] for _ in range(args.cfg5_steps + 4)]
    pf.step(2, scan.range, scan.angle, None, deltas[0], 7)  # clones every touched tile
    first = pf.particle_map_stats()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(1, 1 + args.cfg5_steps):
        pf.step(2, scan.range, scan.angle, None, deltas[k], 7 + k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st, ms = pf.stats(), pf.particle_map_stats()
    ctx.profile_enable(True)
    ctx.profile_read(reset=True)
    ctx.profile_read_map_update(reset=True)
    for k in range(2):
        pf.step(2, scan.range, scan.angle, None, deltas[args.cfg5_steps + 1 + k], 7 + args.cfg5_steps + 1 + k)
    ctx.synchronize()
    ctx.profile_enable(False)
    g_ms, g_launches, g_units = ctx.profile_read(reset=True)
    bpu = BYTES_PER_UNIT["gmapping"]
    k3 = {"bound": "hbm", "achieved": g_units * bpu / (g_ms * 1e-3) / 1e9 if g_ms > 0 else 0.0, "peak": HBM_PEAK_GBS,
          "unit": "GB/s", "kernel": "k_score_gmapping (tile tables)", "bytes_per_unit": bpu, "launches": g_launches,
          "units_launched": g_units, "avg_launch_us": 1e3 * g_ms / max(g_launches, 1), "traffic": None}
    k3["frac"] = k3["achieved"] / HBM_PEAK_GBS
    out = {"metric": "particles/sec at N=%d" % n, "value": n * args.cfg5_steps / dt, "unit": "particles/s",
           "ms_per_step": 1e3 * dt / args.cfg5_steps, "steps": args.cfg5_steps, "n_gpus": 1,
           "workload": "cfg5: GMapping %d particles on 1 GPU, %d beams, %dx%d @%.3f m, per-particle copy-on-write maps, "
                       "area occupancy estimator + blur 0.1 m map update in one batched K6 per step fused behind the "
                       "lock-step likelihood" % (n, scan.n, size, size, scale),
           "roofline": k6_roofline(ctx, "2 extra steps after the timed pass", "cfg5"), "roofline_likelihood": k3,
           "cell_updates_last_step": ms["cell_updates"], "tiles_in_use": ms["tiles_in_use"], "pool_bytes": ms["bytes"],
           "dense_ancestor_bytes": size * size * 48, "cow_copies_first_step": first["cow_copies"],
           "launches_last_step": st["launches"], "scene_build_s": round(t_scene, 1),
           "note": "BASELINE quotes this configuration on 8 GPUs; it fits one MI355X (288 GB); with --gpus N > 1 this "
                   "object is the sharded form (slamhip_gmapping_step_sharded: particles and their maps over the "
                   "ranks, maps migrating over RCCL send / recv on resampling)"}
    pf.close()
    ctx.map_release(2)
    return out


