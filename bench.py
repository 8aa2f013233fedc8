#!/usr/bin/env python3
"""bench.py -- headline benchmark of the scan-matching hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload hc|mc|sweep] [--legs ...]

A "step" is one pass of the hot path over one batch of synthetic input: one
GridScanMatcher::process_scan (the whole accept/reject chain of one scan match) on the
BASELINE.json configuration the metric is quoted on.  Default workload (N=1): configs[1] --
tinySLAM HC scan matcher, 1080-beam scan, 2000x2000 @ 0.05 m occupancy grid, 6-direction
hill-climb with a 128-failed-rounds limit.  The map, the filtered scan and the matcher live in
HBM / on the host before the timed region starts.

value = (scorer calls the reference would make = on_scan_test events) x (filtered beams) / s,
summed over all ranks; speculative evaluations that the replay discards are NOT counted.
For N > 1 single-hypothesis matchers do not shard ("replicas only"): every rank runs
its own independent match, there is no data-path collective, scaling is "weak".  `--gpus N`
without a launcher starts the N ranks itself (torch.distributed.run as a child process).

One JSON line on rank 0 with the extra objects
  roofline        dominant kernel of the headline: algorithmic bytes / HIP-event kernel time vs 8 TB/s
  roofline_valu   what really binds it: VALU issue fraction and instructions per unit (PMC passes,
                  profiles/<tag>_traffic.json) and the measured HBM utilisation
  roofline_sweep  the same scoring arithmetic on flat 4096-pose batches (kernel ceiling)
  cpu_baseline    the compiled reference (oracle/_ref) timed on this box's host cores on a bounded
                  sample of the same workload (rank 0, N=1 only), with an all-cores line for context
  particle_filter BASELINE configs[3] (GMapping, 100 particles sharded over the ranks, RCCL all-gather
                  inside the library) and, on one GPU, the faithful shared-map step and per-particle maps
  cfg5            BASELINE configs[4] on one GPU (500 particles, 8000^2 @ 0.025 m, area estimator + blur,
                  K6 roofline)
  world_loop      one hypothesis scan after scan on the headline's scene: scan upload + match + map update (queued
                  behind the match) per scan, with the CPU restatement's same loop on one core beside it
CPU baselines run BEFORE this process touches the GPU (they use worker processes).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from bench_legs.common import (BYTES_PER_UNIT, HBM_PEAK_GBS, WORKLOADS, compact_line, latency_model,  # noqa: E402
                               load_traffic, roofline_valu, rotating_scenes, sweep_ceiling)
from bench_legs.cpu_baselines import cpu_baseline, mc_reference_baseline, pf_cpu_baselines, world_cpu_baseline  # noqa: E402
from bench_legs.dry_ranks import dry_ranks_main  # noqa: E402
from bench_legs.particle_filter import cfg5_leg, cfg5_sharded_leg, join_shard_group, particle_filter_leg  # noqa: E402
from bench_legs.single_hypothesis import bf_leg, mc_leg, replicas_leg, world_leg  # noqa: E402

ALL_LEGS = ["pf", "pf_update", "pf_maps", "cfg5", "world", "replicas", "bf", "mc", "world_viny"]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="hc", choices=["hc", "mc", "sweep"])
    ap.add_argument("--legs", default=None,
                    help="comma list of %s (default: all on one GPU, pf on several; 'none' = headline only)" % ALL_LEGS)
    ap.add_argument("--size", type=int, default=2000)
    ap.add_argument("--scale", type=float, default=0.05)
    ap.add_argument("--beams", type=int, default=1080)
    ap.add_argument("--sweep-poses", type=int, default=4096)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-procs", type=int, default=0,
                    help="worker processes of the all-cores CPU context lines (0 = one per physical core of the host)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-tail-ab", action="store_true",
                    help="skip the headline's second measurement with SLAMHIP_OPT_INERT_TAIL off (config."
                         "ms_per_step_every_call_scored): profiler runs use it, so that a kernel's dispatches in their trace "
                         "are all of the default path")
    ap.add_argument("--particles", type=int, default=100)
    ap.add_argument("--pf-size", type=int, default=4000)
    ap.add_argument("--pf-steps", type=int, default=10)
    ap.add_argument("--pf-tiles-per-particle", type=int, default=420,
                    help="tile-pool budget per particle of the per-particle-maps leg (768 KiB each)")
    ap.add_argument("--no-pf", action="store_true", help="same as --legs none")
    ap.add_argument("--pf-sigma-xy", type=float, default=0.1, help="slam/particles/sample/xy/sigma (init_gmapping.h:17)")
    ap.add_argument("--pf-sigma-th", type=float, default=0.03, help="slam/particles/sample/theta/sigma (:19-20)")
    ap.add_argument("--pf-maps-sharded", type=int, default=1, choices=[0, 1],
                    help="N > 1 only: the per-particle-maps filter sharded over the ranks, driven through resamplings "
                         "(maps migrate between ranks); 1 = on (default), 0 = the likelihood-only leg alone")
    ap.add_argument("--cfg5-particles", type=int, default=500)
    ap.add_argument("--cfg5-size", type=int, default=8000)
    ap.add_argument("--cfg5-scale", type=float, default=0.025)
    ap.add_argument("--cfg5-steps", type=int, default=8)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend; gloo lets several ranks share one GPU (path testing)")
    ap.add_argument("--strict", action="store_true",
                    help="bit-exact mode (sequential sum + host pose trig) instead of the default")
    ap.add_argument("--seq-sum", action="store_true",
                    help="the reference's beam-order sum (SLAMHIP_SUM_SEQUENTIAL) with device pose trig")
    ap.add_argument("--chain", type=int, default=-1,
                    help="the HC / MC accept chain on the device: 0 off, 1 on, 256/512/1024 = on with that workgroup size")
    ap.add_argument("--chain-mode", type=int, default=-1, choices=[-1, 1, 2],
                    help="device chain of the hill-climbing headline: 1 = a kernel per super-step (csrc/hc_chain.hip), "
                         "2 = one co-resident launch per match (csrc/hc_resident.hip); -1 = the library's default (2)")
    ap.add_argument("--resident-chains", type=int, default=-1, choices=[-1, 0, 1],
                    help="the filter legs' per-particle chains: 1 = one co-resident launch per step where it fits "
                         "(csrc/hc_resident_gm.hip), 0 = a kernel per super-step (csrc/hc_chain.hip); -1 = the library's default (1)")
    ap.add_argument("--tbm-plane", type=int, default=-1, choices=[-1, 0, 1],
                    help="TBM maps: 1 = the 1-cell scorers gather the cell's per-beam probability from the map's 8-byte plane, "
                         "0 = from the 32-byte cell (SLAMHIP_OPT_TBM_PLANE; A/B runs); -1 = the library's default (1)")
    ap.add_argument("--resident-scan", action="store_true",
                    help="headline step = scan_select + process_scan on filtered scans already in HBM (the r01-r03 "
                         "form) instead of the raw scan in (filter + weights + trig + upload inside the step)")
    ap.add_argument("--leg-timeout", type=int, default=480,
                    help="seconds the sharded particle-filter leg may take at N > 1 before the line goes out without it")
    ap.add_argument("--dry-ranks", type=int, default=0,
                    help="no GPU: start this many ranks the way --gpus N does (torch.distributed.run child, rendezvous "
                         "on 127.0.0.1, gloo) and walk the multi-rank control flow of the benchmark on host-only filter "
                         "shards -- block split, all-gathers, identical resampling, the map-migration plan with dummy "
                         "maps point to point, barrier + max-over-ranks timing -- checking every rank against an "
                         "unsharded filter.  So that the first multi-GPU run is not also the first multi-rank run.")
    ap.add_argument("--replica-ks", default="1,2,4,8,16,32,64",
                    help="the `replicas` leg's K values (matches per call)")
    ap.add_argument("--batch", type=int, default=0,
                    help="slamhip_matcher_set_batch on the headline's matcher (Monte Carlo: candidates per super-step, A/B runs)")
    ap.add_argument("--detail-out", default=os.path.join(ROOT, "bench_detail.json"),
                    help="where rank 0 writes the FULL record (every leg's roofline, samples, notes); stdout carries one "
                         "digest line of it, <= 8 KB (the driver keeps the last 8 KB of stdout and parses the last line)")
    ap.add_argument("--no-tie-check", action="store_true",
                    help="default mode without the check of comparisons the tree sum cannot settle")
    args = ap.parse_args()
    if args.no_pf or args.workload == "sweep":
        args.legs = "none"
    if args.legs is None:
        args.leg_set = set(ALL_LEGS) if args.gpus == 1 else {"pf", "cfg5"}
    else:
        args.leg_set = set(x for x in args.legs.split(",") if x and x != "none")
        bad = args.leg_set - set(ALL_LEGS)
        if bad:
            ap.error("unknown leg(s): %s" % sorted(bad))
    return args


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU,
    torch.distributed.run as a CHILD process -- nothing in this process has touched the GPU yet, and it never
    will) and leave with the child's exit code."""
    import socket
    import subprocess
    import torch
    nproc = args.dry_ranks if args.dry_ranks > 0 else args.gpus
    have = torch.cuda.device_count()  # counts devices without initialising the GPU
    if not args.dry_ranks and have < args.gpus and args.backend == "nccl":
        print("bench.py: --gpus %d but only %d GPU(s) visible" % (args.gpus, have), file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.dry_ranks > 0:
        if "WORLD_SIZE" not in os.environ:
            sys.exit(self_launch(args))
        dry_ranks_main(args)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        print("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    import torch
    if torch.cuda.device_count() < 1:
        print("bench.py: no GPU visible; the HIP path has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    from synth import make_scene

    wl = "hc" if args.workload == "sweep" else args.workload
    cell, weighting, kind, params, bkey, desc = WORKLOADS[wl]
    sc_args = dict(cell_model=cell, size=args.size, scale=args.scale, n_beams=args.beams, seed=100 + rank,
                   weighting=weighting)
    sc = make_scene(**sc_args)
    scan = sc["scan"]
    scenes = rotating_scenes(sc, args.beams, weighting) if args.workload != "sweep" else None
    pf_needed = bool(args.leg_set & {"pf", "pf_update", "pf_maps"})
    pf_sc_args = dict(cell_model=2, size=args.pf_size, scale=args.scale, n_beams=args.beams, seed=4)
    pf_sc = make_scene(**pf_sc_args) if pf_needed else None

    # ---- CPU baselines first: worker processes are started while this process is still GPU-free
    cpu_out, pf_cpu_out = None, None
    if world == 1 and rank == 0 and not args.no_cpu and args.workload != "sweep":
        cpu_out = cpu_baseline(sc, sc_args, kind, params, args.cpu_seconds, weighting, args.cpu_procs, scenes)
        if pf_needed:
            pf_cpu_out = pf_cpu_baselines(args, pf_sc, pf_sc_args, min(args.cpu_seconds, 8.0))
    world_cpu_out = None
    if world == 1 and rank == 0 and not args.no_cpu and args.workload != "sweep" and "world" in args.leg_set:
        try:
            world_cpu_out = world_cpu_baseline(sc, kind, params, scenes, weighting)
        except Exception as e:  # noqa: BLE001
            world_cpu_out = {"error": str(e)}

    # BASELINE configs[2] beside the headline: its own scene (TBM cells, viny weights), its own
    # CPU baseline -- the compiled reference's Monte-Carlo matcher on a map the reference's scan adder built
    mc_sc = mc_scenes = mc_cpu = None
    mc_params = WORKLOADS["mc"][3]
    if world == 1 and args.workload == "hc" and not args.strict and not args.seq_sum and (args.leg_set & {"mc", "world_viny"}):
        mc_sc = make_scene(cell_model=1, size=args.size, scale=args.scale, n_beams=args.beams, seed=100 + rank, weighting="viny")
        mc_scenes = rotating_scenes(mc_sc, args.beams, "viny")
        if rank == 0 and not args.no_cpu and "mc" in args.leg_set:
            try:
                mc_cpu = mc_reference_baseline(mc_sc, mc_scenes, mc_params)
            except Exception as e:  # noqa: BLE001
                print("bench.py: Monte-Carlo reference baseline unavailable (%s)" % e, file=sys.stderr)

    # ---- from here on the GPU
    import torch.distributed as dist
    if args.backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()  # ranks may share a GPU under gloo
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":  # RCCL over xGMI
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    args.coll_device = torch.device("cuda", local_rank) if args.backend == "nccl" else torch.device("cpu")

    import __graft_entry__ as ge
    pkg = ge.load_package()
    ctx = pkg.Context(local_rank)
    if args.resident_chains >= 0:
        ctx.set_option(pkg.OPT_RESIDENT_CHAINS, args.resident_chains)
    if args.tbm_plane >= 0:
        ctx.set_option(pkg.OPT_TBM_PLANE, args.tbm_plane)
    ctx.upload_map(0, sc["map"])
    cos_a, sin_a = pkg.beam_trig(scan.angle)
    ctx.scan_upload(scan.range, cos_a, sin_a, scan.weight, scan.factor)
    cfg = pkg.spe_cfg(sum_order=pkg.SUM_SEQUENTIAL, pose_trig=pkg.POSE_TRIG_HOST) if args.strict \
        else (pkg.spe_cfg(sum_order=pkg.SUM_SEQUENTIAL) if args.seq_sum else pkg.spe_cfg())

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    extra = {}
    on_device = False
    m = None
    if args.workload == "sweep":
        # kernel ceiling: flat batch of P device-resident poses, no host round trip
        P = args.sweep_poses
        rs = np.random.RandomState(7 + rank)
        poses = torch.from_numpy(sc["init_pose"] + rs.randn(P, 3) * [0.2, 0.2, 0.1]).cuda()
        scores = torch.empty(P, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()

        def step():
            ctx.score_poses_device(0, cfg, P, poses.data_ptr(), scores.data_ptr())
            return P

        desc = "sweep: %d device-resident poses x %d beams per launch, %s" % (P, scan.n, desc)
    else:
        m = pkg.Matcher(ctx, kind, cfg, params)
        if args.chain >= 0 and kind in ("HC", "MC"):
            m.set_device_chain((args.chain_mode if args.chain_mode > 0 else 2) if args.chain else 0,
                               args.chain if args.chain > 1 else 0)
        elif args.chain_mode > 0 and kind in ("HC", "MC"):
            m.set_device_chain(args.chain_mode)
        if args.no_tie_check:
            m.set_tie_check(0)
        if args.batch > 0:
            m.set_batch(args.batch)
        on_device = kind in ("HC", "MC") and args.chain != 0 and not args.strict
        # the rotating scenes live in HBM before the timed region starts (scan slots); a step selects one (a host
        # pointer swap) and matches it from its own odometry pose
        for j, s_ in enumerate(scenes):
            c_, s__ = pkg.beam_trig(s_["angle"])
            ctx.scan_store(j, s_["range"], c_, s__, s_["weight"])
        beams_of = [s_["range"].size for s_ in scenes]
        step_i = [0]
        evaluated, plain_calls, super_steps = [0], [0], [0]
        closed_calls, closed_units = [0], [0]  # scorer calls (x beams) the chain reported in closed form (the inert tail)
        # The timed step is the reference's process_scan (pose_enumeration_scan_matcher.h:31-77): the RAW scan comes in
        # from host memory, filter_scan (:38), the scan-point weights, the beam trigonometry and the copy to HBM are
        # INSIDE the step (slamhip_scan_filter_upload), then the match.  (`--resident-scan`: the r01-r03 form, the
        # filtered scans already in HBM and a step = scan_select + process_scan; reported either way as
        # config.ms_per_step_resident.)
        raw_upload = [ctx.make_raw_scan(0, s_["raw_range"], s_["raw_angle"], is_occ=s_["is_occ"], weighting=weighting)
                      for s_ in scenes]

        # scorer calls / poses scored of a scene's match: a pure function of the scene (deterministic chains), read
        # from the matcher once per scene before the timed region -- three C calls of bookkeeping per step are not part
        # of process_scan -- and re-checked against the matcher's own counters after it
        per_scene_stats = {}

        def account(k, kept):
            if k not in per_scene_stats or kind != "HC":  # (a Monte-Carlo matcher's engine runs on: no table)
                st_ = m.stats()
                per_scene_stats[k] = (st_["scorer_calls"], st_["poses_evaluated"], st_["launches"], st_["calls_closed_form"])
            c_, e_, l_, t_ = per_scene_stats[k]
            evaluated[0] += e_
            plain_calls[0] += c_
            super_steps[0] += l_
            closed_calls[0] += t_
            closed_units[0] += t_ * kept
            return c_ * kept

        def step_resident():
            k = step_i[0] % len(scenes)
            step_i[0] += 1
            ctx.scan_select(k)
            m.process_scan(0, scenes[k]["init_pose"])
            return account(k, beams_of[k])

        # (r06: ONE C call per step -- slamhip_matcher_process_raw_scan, the reference's process_scan signature: raw scan
        # and initial pose in, pose delta and probability out; r05 made two, slamhip_scan_filter_upload + _process_scan)
        raw_match = [m.make_raw_process_scan(0, s_["raw_range"], s_["raw_angle"], is_occ=s_["is_occ"], weighting=weighting)
                     for s_ in scenes]

        def step_raw():
            k = step_i[0] % len(scenes)
            step_i[0] += 1
            kept, _ = raw_match[k](scenes[k]["init_pose"])
            return account(k, kept)

        step = step_resident if args.resident_scan else step_raw

    barrier()  # (the first torch.cuda.synchronize() initialises torch's own context: not inside the timed region)
    for _ in range(max(args.warmup, len(scenes) if scenes else 0)):
        step()  # (at least one pass over every scene: the chain's run-ahead depth is a running average)
    barrier()
    for _ in range(2):
        step()
    if m is not None:
        step_i[0] = 0
        evaluated[0] = plain_calls[0] = super_steps[0] = 0
        closed_calls[0] = closed_units[0] = 0
    # Pass 1 -- the timed region: exactly K steps, no instrumentation.
    ctx.profile_enable(False)
    barrier()
    t0 = time.perf_counter()
    calls = 0
    step_ms = []
    for _ in range(args.steps):
        ts = time.perf_counter()
        calls += step()
        step_ms.append(1e3 * (time.perf_counter() - ts))
    barrier()
    dt = time.perf_counter() - t0
    if os.environ.get("BENCH_DUMP_STEPS"):
        print("step_ms:", " ".join("%.3f" % x for x in step_ms), file=sys.stderr)
    timed_evaluated, timed_calls = (evaluated[0], plain_calls[0]) if m is not None else (0, 0)
    timed_super_steps = super_steps[0] if m is not None else 0
    timed_closed_calls, timed_closed_units = (closed_calls[0], closed_units[0]) if m is not None else (0, 0)
    units_timed_scored = calls - timed_closed_units  # (matcher steps return calls x beams)
    if m is not None:
        k_last = (step_i[0] - 1) % len(scenes)
        st_chk = m.stats()
        if (st_chk["scorer_calls"], st_chk["poses_evaluated"]) != per_scene_stats[k_last][:2]:
            print("bench.py: the matcher's counters of the last timed match differ from the scene's table", file=sys.stderr)
            sys.exit(6)
    # Pass 2 -- the same K steps again with a HIP event pair attached to every scoring dispatch
    # (stream = the context's own stream): kernel begin..end per launch, for `roofline`.
    ctx.profile_enable(True)
    ctx.profile_read(reset=True)
    barrier()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt_instrumented = time.perf_counter() - t1
    ctx.profile_enable(False)
    k_ms, k_launches, k_units = ctx.profile_read(reset=True)
    kernel_name = "k_score_point"
    parity = None
    if m is not None:
        # the other form of the step, same K steps, for the record
        other = step_raw if args.resident_scan else step_resident
        for _ in range(len(scenes)):
            other()
        barrier()
        t2 = time.perf_counter()
        for _ in range(args.steps):
            other()
        barrier()
        ms_other = 1e3 * (time.perf_counter() - t2) / args.steps
        extra.update(includes_filter_and_upload=not args.resident_scan,
                     **{"ms_per_step_raw_scan_in" if args.resident_scan else "ms_per_step_resident": ms_other})
        if kind == "HC" and timed_closed_calls > 0 and not args.no_tail_ab:
            # the same K steps with SLAMHIP_OPT_INERT_TAIL off: every scorer call of the tail scored, as in r01 - r05
            ctx.set_option(pkg.OPT_INERT_TAIL, 0)
            try:
                for _ in range(len(scenes)):
                    step()
                barrier()
                t3 = time.perf_counter()
                u3 = 0
                for _ in range(args.steps):
                    u3 += step()
                barrier()
                dt3 = time.perf_counter() - t3
                extra["ms_per_step_every_call_scored"] = 1e3 * dt3 / args.steps
                extra["value_every_call_scored"] = float(u3) / dt3  # (what `value` is with the option off)
            finally:
                ctx.set_option(pkg.OPT_INERT_TAIL, 2)
            per_scene_stats.clear()
            for _ in range(len(scenes)):
                step()
        # ---- parity gate on the benchmarked inputs (outside every timed region): the result of every rotating scene
        # through the timed path against what the CPU baseline's reference run returned for the same scene
        want = (cpu_out or {}).pop("_per_scene", None) if rank == 0 else None
        if want and kind == "HC":
            bad, max_rel, kept_bad = [], 0.0, []
            for k_ in sorted(want):
                kept = raw_upload[k_](scenes[k_]["init_pose"])
                r_ = m.process_scan(0, scenes[k_]["init_pose"])
                w_ = want[k_]
                calls_ = m.stats()["scorer_calls"]
                same = calls_ == w_["n_calls"] and [float(x) for x in r_["delta"]] == w_["delta"]
                rel = abs(r_["prob"] / w_["prob"] - 1.0) if w_["prob"] != 0 else abs(r_["prob"])
                max_rel = max(max_rel, rel)
                if kept != w_["filtered_n"]:
                    kept_bad.append(k_)
                if not same or not rel <= 1e-9:
                    bad.append(k_)
            parity = {"scenes": len(want), "traces_equal": len(want) - len(bad), "max_rel_score": max_rel,
                      "filtered_counts_equal": len(want) - len(kept_bad),
                      "against": "the compiled reference's process_scan on the same raw scans (oracle/_ref)"
                                 if cpu_out.get("kind") == "reference" else "the C restatement (oracle/slam_oracle.c)",
                      "what": "scorer calls and pose delta bit for bit, best score within 1e-9 relative (measured above)",
                      "scenes_differing": bad + kept_bad}
        st = m.stats()
        if on_device:
            resident = m.resident_stats()["matches"] > 0
            kernel_name = (("k_hc_chain_resident" if resident else "k_hc_chain_step") if kind == "HC" else
                           ("k_mc_chain_resident" if resident else "k_mc_chain_step"))
            if resident:
                extra.update(resident=m.resident_stats())
        sm = np.sort(np.asarray(step_ms))
        extra.update(scenes="%d rotating (scan, odometry error) pairs resident in HBM: robot poses jittered by N(0, 0.15 m / "
                            "0.04 rad), a fresh range-noise seed each, pose errors 0..3 x (+0.07 m, -0.04 m, +0.03 rad)"
                            % len(scenes),
                     ms_per_match={"min": float(sm[0]), "median": float(np.median(sm)), "max": float(sm[-1])},
                     scorer_calls_per_step=timed_calls / args.steps,
                     scorer_calls_closed_form_per_step=timed_closed_calls / args.steps,
                     closed_form=("the tail of a hill-climbing match: rounds whose candidates can no longer differ from the "
                                  "best pose in any beam's CELL (the steps are below every beam's distance from its cell's "
                                  "edge, certified per root pose with rounding slack; from failed round ~50 on the "
                                  "candidates ARE the best pose bit for bit).  The reference scores them all the same -- "
                                  "the same terms, the same score, a tie, rejected; the chain reports those calls to the "
                                  "observer without scoring them (SLAMHIP_OPT_INERT_TAIL, csrc/hc_resident.hip).  `value` "
                                  "counts them (SURVEY 8d Metric 1: scorer calls = on_scan_test events x beams, as in every "
                                  "round and in cpu_baseline); value_scored_calls_only does not; value_every_call_scored is "
                                  "the same steps with the option off") if timed_closed_calls else None,
                     value_scored_calls_only=(units_timed_scored / dt) if timed_closed_calls else None,
                     poses_evaluated_per_step=timed_evaluated / args.steps,
                     speculation_ratio=timed_evaluated / max(timed_calls, 1),
                     launches_per_step=st["launches"],
                     super_steps_per_match=timed_super_steps / max(args.steps, 1),
                     accept_chain=(("on the device: one process_scan = ONE launch of co-resident workgroups that exchange "
                                    "their scores inside it and replay every super-step's speculation tree "
                                    "(csrc/hc_resident.hip, csrc/mc_resident.hip)") if resident else
                                   ("on the device: one process_scan = a chain of kernels, each replaying the previous "
                                    "one's speculation tree (csrc/hc_chain.hip, csrc/mc_chain.hip)")) if on_device else
                                  "on the host: speculative batches, replay between launches",
                     kernel_busy_frac=k_ms / (1e3 * dt_instrumented) if dt_instrumented > 0 else None,
                     host_us_last_step={k: round(st[k], 1) for k in ("build_us", "stage_us", "score_us", "replay_us")})

    units = float(calls) * scan.n if m is None else float(calls)  # (matcher steps return calls x their scan's beams)
    t_max, units_all = dt, units
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=args.coll_device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        uu = torch.tensor([units], dtype=torch.float64, device=args.coll_device)
        dist.all_reduce(uu, op=dist.ReduceOp.SUM)
        t_max, units_all = tt.item(), uu.item()

    ceiling = None
    if rank == 0 and args.workload != "sweep" and not args.strict:  # outside the timed region
        ceiling = sweep_ceiling(pkg, ctx, pkg.spe_cfg(), sc, scan.n, args.sweep_poses, 50, BYTES_PER_UNIT[bkey], torch)
    if m is not None:
        m.close()

    world_out = None  # (the world-loop leg, filled in below; emit_line reads it when the line goes out)
    replicas_out = None
    extra_legs = {}

    # devices the ranks really run on (under `--backend gloo` several ranks may share one: that is not N GPUs)
    n_devices = world
    if world > 1:
        ids = [None] * world
        try:
            dist.all_gather_object(ids, (os.uname().nodename, local_rank))
            n_devices = len(set(ids))
        except Exception:  # noqa: BLE001
            n_devices = min(world, torch.cuda.device_count())

    def emit_line(pf_out, cfg5_out):
        """rank 0's ONE JSON line (the headline is complete before the secondary legs start)"""
        bpu = BYTES_PER_UNIT[bkey]
        achieved = (k_units * bpu) / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        traffic, traffic_src = load_traffic(args.workload)
        avg_us = 1e3 * k_ms / max(k_launches, 1)
        # the same kernel time priced with the REFERENCE's scorer calls only (speculative poses the replay discards
        # carry no useful bytes): units of the timed pass = the instrumented pass's (same K steps, same scenes)
        useful = (units * bpu) / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        out = {
            "metric": "pose-candidates*beams/sec (1080-beam scan, 2000^2 grid)",
            "value": units_all / t_max,
            "unit": "pose-candidates*beams/s",
            "n_gpus": n_devices,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * t_max / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": desc,
                       "beams_after_filter": scan.n if scenes is None else float(np.mean([x["range"].size for x in scenes])),
                       "mode": "strict (sequential sum, host trig)" if args.strict else
                               ("beam-order sum, device sincos" if args.seq_sum else
                                ("default without the tie check (canonical tree sum, device sincos)" if args.no_tie_check else
                                 "default (canonical tree sum, device sincos; comparisons the tree sum cannot "
                                 "settle decided from beam-order sums)")),
                       "parallelism": "replicas x%d (no collective)" % world if world > 1 else "1 gpu",
                       "ranks": world,
                       "backend": args.backend if world > 1 else None,
                       **extra},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "achieved_useful": useful, "frac_useful": useful / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": kernel_name, "bytes_per_unit": bpu,
                         "launches": k_launches, "units_launched": k_units,
                         "avg_launch_us": avg_us,
                         "timing": "HIP events attached to each %s dispatch on the context's stream, second pass of "
                                   "the same %d steps (%.4f ms/step with the events attached; the timed pass carries "
                                   "none)" % (kernel_name, args.steps, 1e3 * dt_instrumented / args.steps)},
        }
        rv = roofline_valu(args.workload, avg_us)
        if rv:
            out["roofline_valu"] = rv
        if on_device and kind == "HC" and "ms_per_match" in extra:
            lm = latency_model(extra["ms_per_match"]["median"], extra.get("super_steps_per_match"),
                               resident=kernel_name == "k_hc_chain_resident")
            if lm:
                out["latency_model"] = lm
        if ceiling is not None:
            out["roofline_sweep"] = ceiling
        if cpu_out is not None:
            cpu_out.pop("_per_scene", None)
            out["cpu_baseline"] = cpu_out
        out["parity"] = parity if parity is not None else {
            "scenes": 0, "note": "not checked in this run: " + ("--no-cpu" if args.no_cpu else (
                "the gate runs on rank 0 of a 1-GPU hill-climbing run, where the CPU baseline has the reference's "
                "results of the benchmarked scenes"))}
        if pf_out is not None:
            out["particle_filter"] = pf_out
            if pf_cpu_out:
                # like beside like (VERDICT r5 "What's weak" 5): `value` is the likelihood step alone -- its pair is the
                # reference filter with its scan adder switched off; the reference's FULL step (map update inside) is
                # the pair of `with_map_update`
                lk = pf_cpu_out.get("likelihood_only") if isinstance(pf_cpu_out, dict) else None
                full = {k_: v_ for k_, v_ in pf_cpu_out.items() if k_ != "likelihood_only"} if isinstance(pf_cpu_out, dict) else pf_cpu_out
                if isinstance(lk, dict) and "value" in lk:
                    pf_out["cpu_baseline"] = lk
                    pf_out["cpu_baseline_full_step"] = full
                else:
                    pf_out["cpu_baseline"] = full
                if isinstance(pf_out.get("with_map_update"), dict) and "error" not in pf_out["with_map_update"]:
                    pf_out["with_map_update"]["cpu_baseline"] = full
        if cfg5_out is not None:
            out["cfg5"] = cfg5_out
        if world_out is not None:
            out["world_loop"] = world_out
        if replicas_out is not None:
            out["replicas"] = replicas_out
        for k_, v_ in extra_legs.items():
            if v_ is not None:
                out[k_] = v_
        # the full record goes to the sidecar; stdout gets ONE digest line of it (<= 8 KB: bench_legs/common.py)
        detail = None
        try:
            with open(args.detail_out, "w") as f:
                json.dump(out, f)
                f.write("\n")
            detail = os.path.relpath(args.detail_out, ROOT) if args.detail_out.startswith(ROOT) else args.detail_out
        except OSError as e:
            print("bench.py: cannot write %s (%s)" % (args.detail_out, e), file=sys.stderr)
        line = compact_line(out, detail)
        # RCCL's banner sits in C stdio's buffer when stdout is a pipe: push it out first, so that the line is the last one
        import ctypes
        ctypes.CDLL(None).fflush(None)
        sys.stderr.flush()
        print(line, flush=True)

    # The secondary legs run AFTER the headline is complete.  With more than one rank the particle-filter leg joins
    # an RCCL group inside the library: should that ever block (a fabric problem is not this benchmark's to sit
    # out), a watchdog prints the headline with the failure noted and ends the process, so the driver still gets
    # its line.
    watchdog = None
    if world > 1 and (pf_needed or "cfg5" in args.leg_set):
        import threading

        def give_up():
            if rank == 0:
                emit_line({"error": "the sharded particle-filter leg did not finish within %d s" % args.leg_timeout}, None)
                sys.stdout.flush()
            os._exit(5)  # the line is out, the run still failed: a hung leg is not a success

        watchdog = threading.Timer(args.leg_timeout, give_up)
        watchdog.daemon = True
        watchdog.start()
    pf_out = None
    if pf_needed:
        try:
            pf_out = particle_filter_leg(args, pkg, ctx, pf_sc, rank, world, dist, torch)
        except Exception as e:  # noqa: BLE001  (the headline line must still go out; the failure is reported in it)
            import traceback
            traceback.print_exc()
            pf_out = {"error": "%s: %s" % (type(e).__name__, e)}
    # ... and its WEAK-scaling form: `--particles` particles PER rank.  A filter step is a latency
    # chain whose length does not shrink with the shard, so the strong curve flattens early by construction; the weak
    # one shows what more GPUs do buy -- more particles in the same step time.  One GPU: the model of it.
    if pf_out is not None and "error" not in pf_out and "pf" in args.leg_set and "value" in pf_out:
        try:
            if world > 1:
                import copy
                wargs = copy.copy(args)
                wargs.particles = args.particles * world
                wargs.leg_set = {"pf"}
                wargs.pf_maps_sharded = 0
                w_ = particle_filter_leg(wargs, pkg, ctx, pf_sc, rank, world, dist, torch)
                args._joined = getattr(wargs, "_joined", getattr(args, "_joined", False))
                pf_out["weak"] = {"scaling": "weak", "ranks": world, "particles": wargs.particles,
                                  "particles_per_rank": args.particles, "unit": "particles/s",
                                  **{k_: w_[k_] for k_ in ("value", "ms_per_step", "steps", "resamplings", "collective", "error",
                                                           "skipped") if k_ in w_}}
            else:
                coll_us = (pf_out.get("scaling_model") or {}).get("collective_us_one_rank_group") or 0.0
                t1_ = pf_out["ms_per_step"]
                pf_out["weak_scaling_model"] = {
                    "by_ranks": [{"ranks": G, "particles": G * args.particles,
                                  "predicted_ms_per_step": t1_ + (coll_us * 1e-3 if G > 1 else 0.0),
                                  "predicted_particles_per_s": G * args.particles / ((t1_ + (coll_us * 1e-3 if G > 1 else 0.0)) * 1e-3)}
                                 for G in (1, 2, 4, 8)],
                    "note": "%d particles per GPU: every rank runs this GPU's step, plus the step's one all-gather of "
                            "G x %d raw weights (measured on a 1-rank group here; more ranks add link latency)"
                            % (args.particles, args.particles)}
        except Exception as e:  # noqa: BLE001  (the line must still go out)
            import traceback
            traceback.print_exc()
            pf_out["weak"] = {"error": "%s: %s" % (type(e).__name__, e)}
    cfg5_out = None
    if "cfg5" in args.leg_set and world == 1:
        try:
            cfg5_out = cfg5_leg(args, pkg, ctx, torch)
        except pkg.SlamHipError as e:
            cfg5_out = {"error": str(e)}
    elif "cfg5" in args.leg_set:
        try:
            if args.backend == "nccl":
                join_shard_group(args, pkg, ctx, rank, world, dist, torch)
            cfg5_out = cfg5_sharded_leg(args, pkg, ctx, rank, world, dist, torch)
        except Exception as e:  # noqa: BLE001  (the line must still go out)
            import traceback
            traceback.print_exc()
            cfg5_out = {"error": "%s: %s" % (type(e).__name__, e)}

    if "world" in args.leg_set and world == 1 and args.workload != "sweep":
        try:
            world_out = world_leg(args, pkg, ctx, sc, cfg, kind, params, scenes)
            if world_cpu_out:
                world_out["cpu_baseline"] = world_cpu_out
        except pkg.SlamHipError as e:
            world_out = {"error": str(e)}

    bf_out = None
    if "bf" in args.leg_set and world == 1 and args.workload == "hc" and not args.strict and not args.seq_sum:
        try:
            bf_out = bf_leg(args, pkg, ctx, sc, scenes, BYTES_PER_UNIT[bkey], ceiling)
        except pkg.SlamHipError as e:
            bf_out = {"error": str(e)}
    extra_legs["brute_force"] = bf_out

    if "replicas" in args.leg_set and world == 1 and args.workload == "hc" and not args.strict and not args.seq_sum:
        try:
            replicas_out = replicas_leg(args, pkg, ctx, cfg, params, scenes, BYTES_PER_UNIT[bkey],
                                        ks=tuple(int(v) for v in args.replica_ks.split(",")))
        except pkg.SlamHipError as e:
            replicas_out = {"error": str(e)}

    mc_parity = None
    if mc_sc is not None and "mc" in args.leg_set:
        try:
            mc_out = mc_leg(args, pkg, ctx, mc_sc, mc_scenes, mc_params, mc_cpu)
            mc_parity = mc_out.get("parity")
            if mc_cpu:
                mc_out["cpu_baseline"] = {k_: v_ for k_, v_ in mc_cpu.items() if not k_.startswith("_")}
        except pkg.SlamHipError as e:
            mc_out = {"error": str(e)}
        extra_legs["monte_carlo"] = mc_out
    if mc_sc is not None and "world_viny" in args.leg_set:
        try:
            extra_legs["world_loop_viny"] = world_leg(args, pkg, ctx, mc_sc, pkg.spe_cfg(), "MC", mc_params, mc_scenes,
                                                      preset="viny")
        except pkg.SlamHipError as e:
            extra_legs["world_loop_viny"] = {"error": str(e)}

    if watchdog is not None:
        watchdog.cancel()
    if rank == 0:
        emit_line(pf_out, cfg5_out)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()
    if mc_parity is not None and (mc_parity["traces_equal"] != mc_parity["scenes"] or
                                  mc_parity["filtered_counts_equal"] != mc_parity["scenes"]):
        print("bench.py: PARITY FAILURE of the Monte-Carlo leg on scenes %r" % mc_parity["scenes_differing"], file=sys.stderr)
        sys.exit(4)
    if parity is not None and (parity["traces_equal"] != parity["scenes"] or parity["filtered_counts_equal"] != parity["scenes"]):
        print("bench.py: PARITY FAILURE on the benchmarked scenes %r (the line above carries the details)"
              % parity["scenes_differing"], file=sys.stderr)
        sys.exit(4)


if __name__ == "__main__":
    main()

