"""bench_legs.single_hypothesis -- the legs around the headline's matcher: the world loop (scan upload + match + map
update per scan), K matches per call, the brute-force sweep."""
import time

import numpy as np

from .common import BYTES_PER_UNIT, HBM_PEAK_GBS, load_traffic


def world_leg(args, pkg, ctx, sc, cfg, kind, params, scenes, preset="tiny", map_data=None):
    """One hypothesis, scan after scan, everything through the C-ABI: upload the (filtered) scan, match from the
    odometry pose, append the scan to the map from the matched pose -- SingleStateHypothesisLaserScanGridWorld::
    handle_observation (single_state_hypothesis_laser_scan_grid_world.h:52-65) with the map resident in HBM and its
    update queued behind the match (slamhip_map_set_deferred), as host/slamhip_resident_world.h runs it -- over the
    rotating scans (every scan arrives from the host, as a sensor's would).  preset "tiny": MeanProbabilityCell map,
    the headline's matcher; "viny": TBM cells, viny weights, Monte Carlo, the scan adder of viny_slam_base.properties.
    Parity of this loop against the reference's world: tests/test_gpu_world.py."""
    m0 = map_data if map_data is not None else sc["map"]
    trig = [pkg.beam_trig(s["angle"]) for s in scenes]
    ctx.map_bind(5, m0.cell_model, m0.width, m0.height, m0.origin, m0.scale, m0.unknown)
    ctx.map_upload_window(5, 0, 0, m0.payload)
    ctx.map_set_auto_grow(5, True)
    m = pkg.Matcher(ctx, kind, cfg, params)
    ctx.map_set_deferred(True)
    it = [0]
    viny = preset == "viny"
    rule = pkg.RULE_TBM if viny else pkg.RULE_MEAN
    adder = dict(quality=0.9, base=(0.95, 0.04, 0.01, 0.003), blur=0.3) if viny else {}

    # (argument blocks made once per scan: what a caller that holds its scans in C arrays passes -- r06; the three calls
    # and what they do are unchanged)
    uploads = [ctx.make_scan_upload(s["range"], trig[k][0], trig[k][1], s["weight"]) for k, s in enumerate(scenes)]
    appends = [ctx.make_map_append_scan(5, rule, s["range"], trig[k][0], trig[k][1], **adder) for k, s in enumerate(scenes)]

    def one_scan():
        k = it[0] % len(scenes)
        it[0] += 1
        s = scenes[k]
        uploads[k]()
        r = m.process_scan(5, s["init_pose"])
        appends[k](s["init_pose"] + r["delta"])

    for _ in range(len(scenes)):
        one_scan()
    updates = ctx.map_drain()
    n = 2 * len(scenes) if viny else max(2 * len(scenes), args.steps)
    ctx.synchronize()
    per = []
    t0 = time.perf_counter()
    for _ in range(n):
        ts = time.perf_counter()
        one_scan()
        per.append(1e3 * (time.perf_counter() - ts))
    updates = ctx.map_drain()
    dt = time.perf_counter() - t0
    ctx.map_set_deferred(False)
    m.close()
    ctx.map_release(5)
    per = np.sort(np.asarray(per))
    return {"metric": "scans/s, one hypothesis: scan upload + match + map update per scan", "value": n / dt,
            "unit": "scans/s", "ms_per_scan": 1e3 * dt / n, "scans": n, "cell_updates_per_scan": updates / n,
            "ms_per_scan_host_side": {"min": float(per[0]), "median": float(np.median(per)), "max": float(per[-1])},
            "preset": "vinySLAM: TBM cells, viny weights, MC %s, const estimator 0.95 / 0.04, 0.01 / 0.003, blur 0.3" % params
                      if viny else "tinySLAM: MeanProbabilityCell, even weights, HC %s" % params,
            "note": "map resident in HBM, the update queued behind the match on the context's "
                    "stream and drained at the end of the timed region; %d rotating (scan, odometry error) pairs, each "
                    "scan uploaded from the host inside its step" % len(scenes)}


def mc_leg(args, pkg, ctx, sc, scenes, params, cpu_mc, matches=32):
    """BASELINE configs[2] (vinySLAM MC: 4096 Monte-Carlo candidates, 1080 beams, TBM cells, viny weights, 2000^2 grid)
    the way the headline runs cfg2: a step = the reference's process_scan on a RAW scan from host memory (filter, viny
    weights, beam trig, upload, then ONE co-resident launch: csrc/mc_resident.hip).  `matches` timed steps on one
    matcher (its engine runs on from match to match, like the reference's), a second pass with HIP events on every
    dispatch for `roofline`, and -- where the CPU baseline ran -- a parity gate: every benchmarked scene through a
    FRESH matcher against what the compiled reference returned for it (scorer calls and delta bit for bit)."""
    from synth import MapData
    m0 = sc["map"]
    if cpu_mc and cpu_mc.get("_map"):
        g = cpu_mc["_map"]  # the map the reference's scan adder built (the parity gate needs the very same cells)
        m0 = MapData(1, g["payload"], g["origin"], g["scale"], g["unknown"])
    ctx.upload_map(6, m0)
    raw = [ctx.make_raw_scan(6, s_["raw_range"], s_["raw_angle"], is_occ=s_["is_occ"], weighting="viny") for s_ in scenes]
    m = pkg.Matcher(ctx, "MC", pkg.spe_cfg(), params)
    if args.chain_mode > 0:
        m.set_device_chain(args.chain_mode)
    it = [0]
    tot = dict(calls=0, units=0, evaluated=0, steps=0)

    def step(count=True):
        k = it[0] % len(scenes)
        it[0] += 1
        kept = raw[k](scenes[k]["init_pose"])
        m.process_scan(6, scenes[k]["init_pose"])
        if count:
            st = m.stats()
            tot["calls"] += st["scorer_calls"]
            tot["units"] += st["scorer_calls"] * kept
            tot["evaluated"] += st["poses_evaluated"]
            tot["steps"] += st["launches"]

    for _ in range(len(scenes)):
        step(False)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(matches):
        step()
    ctx.synchronize()
    dt = time.perf_counter() - t0
    ctx.profile_enable(True)
    ctx.profile_read(reset=True)
    for _ in range(matches):
        step(False)
    ctx.synchronize()
    ctx.profile_enable(False)
    k_ms, k_launches, k_units = ctx.profile_read(reset=True)
    res = m.resident_stats()
    m.close()
    bpu = BYTES_PER_UNIT["tbm"]
    achieved = k_units * bpu / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    traffic, traffic_src = load_traffic("mc")
    out = {"metric": "pose-candidates*beams/sec (1080-beam scan, 2000^2 grid), Monte Carlo",
           "workload": "cfg3: vinySLAM MC(seed %d, %d attempts), 1080 beams, TBM cell, viny weights, 2000x2000 @0.05 m, raw "
                       "scan in" % (params[0], params[4]),
           "value": tot["units"] / dt, "unit": "pose-candidates*beams/s", "ms_per_step": 1e3 * dt / matches, "steps": matches,
           "scorer_calls_per_step": tot["calls"] / matches, "speculation_ratio": tot["evaluated"] / max(tot["calls"], 1),
           "super_steps_per_match": tot["steps"] / matches, "resident": res,
           "map": "built by the compiled reference's scan adder (the CPU baseline's map)" if m0 is not sc["map"] else "synthetic",
           "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "traffic": traffic, "traffic_source": traffic_src,
                        "kernel": "k_mc_chain_resident" if res["matches"] > res["gave_up"] else "k_mc_chain_step",
                        "bytes_per_unit": bpu, "launches": k_launches, "units_launched": k_units,
                        "avg_launch_us": 1e3 * k_ms / max(k_launches, 1),
                        "timing": "HIP events attached to each dispatch, second pass of the same %d steps" % matches}}
    # what a (pose, beam) really gathers since r06: the cell's per-beam probability from the map's 8-byte plane
    # (DeviceMap::d_prob; SLAMHIP_OPT_TBM_PLANE) instead of the 32-byte belief -- `bytes_per_unit` stays SURVEY 8d's 56
    plane_on = bool(ctx.get_option(pkg.OPT_TBM_PLANE))
    gather = 16 + 8 + (8 if plane_on else 32)  # beam record + weight + cell
    out["roofline"].update(probability_plane=plane_on, gather_bytes_per_unit=gather,
                           achieved_gather_gbs=(k_units * gather / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0))
    if cpu_mc and cpu_mc.get("_per_scene"):
        want = cpu_mc["_per_scene"]
        bad, kept_bad, max_rel = [], [], 0.0
        for k_ in sorted(want):
            f = pkg.Matcher(ctx, "MC", pkg.spe_cfg(), params)
            if args.chain_mode > 0:
                f.set_device_chain(args.chain_mode)
            kept = raw[k_](scenes[k_]["init_pose"])
            r_ = f.process_scan(6, scenes[k_]["init_pose"])
            calls_ = f.stats()["scorer_calls"]
            f.close()
            w_ = want[k_]
            rel = abs(r_["prob"] / w_["prob"] - 1.0) if w_["prob"] != 0 else abs(r_["prob"])
            max_rel = max(max_rel, rel)
            if kept != w_["filtered_n"]:
                kept_bad.append(k_)
            if calls_ != w_["n_calls"] or [float(x) for x in r_["delta"]] != w_["delta"] or not rel <= 1e-9:
                bad.append(k_)
        out["parity"] = {"scenes": len(want), "traces_equal": len(want) - len(bad), "max_rel_score": max_rel,
                         "filtered_counts_equal": len(want) - len(kept_bad), "scenes_differing": bad + kept_bad,
                         "against": "the compiled reference's MonteCarloScanMatcher::process_scan on the same raw scans and "
                                    "the same map, a fresh matcher (same seed) per scene on both sides",
                         "what": "scorer calls and pose delta bit for bit, best score within 1e-9 relative"}
    ctx.map_release(6)
    return out


def replicas_leg(args, pkg, ctx, cfg, params, scenes, bpu, ks=(1, 2, 4, 8, 16, 32, 64)):
    """K independent matches per call (slamhip_matcher_process_scan_batch; SURVEY 8e: single-hypothesis matchers
    replicate, they do not shard): the headline's matcher on K of the rotating scenes at once, all K accept chains
    advancing in shared launches (grid.y = match).  Scans are resident in HBM (the slots the headline stored); K = 1
    is the lone-match latency.  Per K: whole-call throughput in the headline's unit (reference-equivalent scorer
    calls x beams / s), matches/s, ms per call, the chain kernel's roofline over the calls of a second, instrumented
    pass (HIP events on every dispatch), and the speculation ratio."""
    out = []
    n_sc = len(scenes)
    beams = [s["range"].size for s in scenes]
    for K in ks:
        m = pkg.Matcher(ctx, "HC", cfg, params)
        if args.chain_mode > 0:
            m.set_device_chain(args.chain_mode)
        groups = [[(g * K + j) % n_sc for j in range(K)] for g in range(max(1, n_sc // K) if K <= n_sc else 1)]
        blocks = [m.make_batch([dict(map_id=0, scan_slot=k, init_pose=scenes[k]["init_pose"]) for k in grp])
                  for grp in groups]
        calls_per_block = []
        for blk in blocks:  # warm-up: every block twice (tree shapes, run-ahead depth)
            m.process_scan_batch(blk)
            m.process_scan_batch(blk)
            calls_per_block.append([m.batch_stats(j) for j in range(K)])
        n_calls = max(len(blocks), 32)  # (the leg's own sample: at least 32 calls per K whatever --steps says)
        ctx.synchronize()
        t0 = time.perf_counter()
        units = evaluated = plain = 0
        for it in range(n_calls):
            g = it % len(blocks)
            m.process_scan_batch(blocks[g])
            for j, st in enumerate(calls_per_block[g]):
                units += st["scorer_calls"] * beams[groups[g][j]]
                plain += st["scorer_calls"]
                evaluated += st["poses_evaluated"]
        ctx.synchronize()
        dt = time.perf_counter() - t0
        ctx.profile_enable(True)
        ctx.profile_read(reset=True)
        for it in range(n_calls):
            m.process_scan_batch(blocks[it % len(blocks)])
        ctx.synchronize()
        ctx.profile_enable(False)
        k_ms, k_launches, k_units = ctx.profile_read(reset=True)
        achieved = k_units * bpu / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        st = m.stats()
        on_chain = sum(1 for x in calls_per_block[0] if x["on_device_chain"])
        res = m.resident_stats()
        out.append({"K": K, "co_resident_launches": res["matches"], "co_resident_gave_up": res["gave_up"], "value": units / dt, "unit": "pose-candidates*beams/s", "matches_per_s": n_calls * K / dt,
                    "ms_per_call": 1e3 * dt / n_calls, "calls": n_calls,
                    "speculation_ratio": evaluated / max(plain, 1), "kernels_per_call": st["kernels_launched"],
                    "super_steps_longest_chain": st["launches"], "matches_on_shared_launches": on_chain,
                    "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": achieved / HBM_PEAK_GBS,
                                 "kernel": "k_hc_chain_resident" if res["matches"] > res["gave_up"] else "k_hc_chain_step",
                                 "bytes_per_unit": bpu,
                                 "launches": k_launches, "avg_launch_us": 1e3 * k_ms / max(k_launches, 1),
                                 "kernel_busy_frac": None}})
        m.close()
    return {"metric": "pose-candidates*beams/sec, K independent matches per call (slamhip_matcher_process_scan_batch)",
            "by_K": out,
            "note": "the same matcher and rotating scenes as the headline; a match of a batch returns the trace of its "
                    "lone run bit for bit (tests/test_gpu_batch.py)"}


def bf_leg(args, pkg, ctx, sc, scenes, bpu, ceiling):
    """The brute-force matcher on the search-space evaluator's sweep -- 201 x 201 poses around the odometry pose
    (src/utils/pose2D_search_space_evaluator.cpp:154-184) -- as ONE flat K1 launch + a device arg-max
    (csrc/bf_device.hip) on the headline's map and rotating scenes (filtered scans resident in HBM): whole
    process_scan calls per second in the headline's unit, beside the flat sweep's kernel rate (roofline_sweep)."""
    rng9 = [-0.5, 0.5 - 1e-9, 0.005, -0.5, 0.5 - 1e-9, 0.005, 0.0, 0.0, 1.0]  # 201 x 201 x 1
    m = pkg.Matcher(ctx, "BF", pkg.spe_cfg(), rng9)
    beams = [s["range"].size for s in scenes]
    for k in range(3):
        ctx.scan_select(k)
        m.process_scan(0, scenes[k]["init_pose"])
    n_calls = 24
    ctx.synchronize()
    t0 = time.perf_counter()
    units = calls = 0
    for it in range(n_calls):
        k = it % len(scenes)
        ctx.scan_select(k)
        m.process_scan(0, scenes[k]["init_pose"])
        st = m.stats()
        units += st["scorer_calls"] * beams[k]
        calls += st["scorer_calls"]
    ctx.synchronize()
    dt = time.perf_counter() - t0
    st = m.stats()
    m.close()
    out = {"metric": "pose-candidates*beams/sec, brute-force matcher, 201 x 201 search space per process_scan",
           "value": units / dt, "unit": "pose-candidates*beams/s", "ms_per_match": 1e3 * dt / n_calls,
           "poses_per_match": calls / n_calls, "kernels_per_match": st["kernels_launched"],
           "on_device": st["kernels_launched"] == 4,
           "roofline": {"bound": "hbm", "achieved": units / dt * bpu / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": units / dt * bpu / 1e9 / HBM_PEAK_GBS, "kernel": "k_score_point", "bytes_per_unit": bpu,
                        "timing": "whole process_scan calls (pose list + sweep + arg-max + result over PCIe), host clock"}}
    if ceiling and ceiling.get("achieved"):
        out["fraction_of_flat_sweep_rate"] = out["roofline"]["achieved"] / ceiling["achieved"]
    return out


