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

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
BYTES_PER_UNIT = {"occ": 24, "tbm": 56, "gmapping": 232}  # SURVEY 8d algorithmic bytes / (pose, beam)
K6_BYTES_PER_RECORD = 64  # per (beam, cell) 2 x 32 B read-modify-write
ALL_LEGS = ["pf", "pf_update", "pf_maps", "cfg5", "world", "replicas", "bf"]

WORKLOADS = {
    # name: (cell model, weighting, matcher kind, params, bytes key, description)
    "hc": (0, "even", "HC", [128, 0.1, 0.1], "occ",
           "cfg2: tinySLAM HC(dt 0.1, dr 0.1, failed-rounds 128), 1080 beams, 2000x2000 @0.05 m, occupancy cell"),
    "mc": (1, "viny", "MC", [666666, 0.2, 0.1, 4096, 4096], "tbm",
           "cfg3: vinySLAM MC(seed 666666, 4096 attempts), 1080 beams, TBM cell, viny weights, 2000x2000 @0.05 m"),
}


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
    ap.add_argument("--particles", type=int, default=100)
    ap.add_argument("--pf-size", type=int, default=4000)
    ap.add_argument("--pf-steps", type=int, default=10)
    ap.add_argument("--pf-tiles-per-particle", type=int, default=420,
                    help="tile-pool budget per particle of the per-particle-maps leg (768 KiB each)")
    ap.add_argument("--no-pf", action="store_true", help="same as --legs none")
    ap.add_argument("--pf-sigma-xy", type=float, default=0.1, help="slam/particles/sample/xy/sigma (init_gmapping.h:17)")
    ap.add_argument("--pf-sigma-th", type=float, default=0.03, help="slam/particles/sample/theta/sigma (:19-20)")
    ap.add_argument("--pf-maps-sharded", action="store_true",
                    help="N > 1 only: also run the per-particle-maps filter sharded over the ranks (maps "
                         "migrate between ranks on resampling)")
    ap.add_argument("--cfg5-particles", type=int, default=500)
    ap.add_argument("--cfg5-size", type=int, default=8000)
    ap.add_argument("--cfg5-scale", type=float, default=0.025)
    ap.add_argument("--cfg5-steps", type=int, default=4)
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
    ap.add_argument("--batch", type=int, default=0,
                    help="slamhip_matcher_set_batch on the headline's matcher (Monte Carlo: candidates per super-step, A/B runs)")
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


def load_profile_json():
    """The newest committed PMC summary (profiles/<tag>_traffic.json, written by tools/summarize_profiles.py
    from separate rocprofv3 --pmc passes of these same commands).  PMC counters cannot be collected from
    inside the benchmark, so these are the roofline fields not measured live."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not files:
        return None, None
    return json.load(open(files[-1])), os.path.relpath(files[-1], ROOT)


def load_traffic(workload):
    d, path = load_profile_json()
    w = (d or {}).get("workloads", {}).get(workload)
    if not w or "bytes_per_launch" not in w:
        return None, None
    return w["bytes_per_launch"], "%s (%s)" % (path, w["correction"])


def roofline_valu(workload, avg_launch_us):
    """The bound that actually binds: the cell gathers are cache hits, so HBM idles and the
    kernels are limited by VALU issue (FP64 and integer instructions alike take four cycles per wave64 on a
    16-lane SIMD).  From the SQ counters of the committed PMC passes and the live kernel time."""
    d, path = load_profile_json()
    w = (d or {}).get("workloads", {}).get(workload)
    if not w or "valu" not in w:
        return None
    v = dict(w["valu"])
    out = {"bound": "valu", "source": path, "kernel": w.get("kernel"),
           "valu_wave_insts_per_launch": v.get("SQ_INSTS_VALU"),
           "valu_insts_per_unit": v.get("insts_per_unit"),
           "valu_issue_frac": v.get("issue_frac"),
           "peak": "1024 SIMDs x 1 VALU wave-instruction per 4 cycles",
           "note": v.get("note")}
    if "bytes_per_launch" in w and avg_launch_us:
        out["hbm_gbs_measured"] = w["bytes_per_launch"] / (avg_launch_us * 1e-6) / 1e9
        out["hbm_utilisation"] = out["hbm_gbs_measured"] / HBM_PEAK_GBS
    return out


# the dependent chain of ONE super-step of the device-resident hill-climbing chain (1024-thread workgroups,
# point OOPE), priced from the guide's primitive latencies at ~2.1 GHz (global_load: L2 hit 200 cycles, memory 900; a
# dependent FP64 / integer VALU op 8 cycles; LDS read ~64 cycles; kernel boundary 1.45 us), beside the wall_clock64
# stamps of tools/hc_chain_stamps.py (profiles/r03_chain_stamps.txt).
HC_LATENCY_MODEL_US = {"boundary": 1.45, "staged": 0.90, "replayed": 0.52, "pose": 0.50, "terms": 0.25, "stored": 0.20}
HC_LATENCY_STAMPS_US = {"boundary": 1.85, "staged": 1.61, "replayed": 1.88, "pose": 0.69, "terms": 0.93, "stored": 0.89}
# r04, the co-resident chain (csrc/hc_resident.hip): no kernel boundary and no staging -- the scores cross the chip as
# granules: one write-through store, one hop (MI355X_MICROARCH.md handoff-1to1, idle: 0.8 us) and half a poll period;
# stamps: tools/hc_resident_stamps.py (profiles/r04_resident_stamps.txt)
HC_RESIDENT_MODEL_US = {"gather": 1.05, "replayed": 0.52, "pose": 0.50, "terms": 0.25, "sum_publish": 0.30}
HC_RESIDENT_STAMPS_US = {"gather": 1.51, "replayed": 1.69, "pose": 0.67, "terms": 1.07, "sum_publish": 0.91, "loop": 0.32}


def latency_model(ms_per_match, super_steps, resident=False):
    """achieved / model for the headline's real bound, the serial accept chain: a match is `super_steps` super-steps
    in a row, each a chain of dependent memory round trips, barriers and FP64 sequences that no amount of width
    shortens."""
    if not super_steps or not ms_per_match:
        return None
    stages, stamps = (HC_RESIDENT_MODEL_US, HC_RESIDENT_STAMPS_US) if resident else (HC_LATENCY_MODEL_US, HC_LATENCY_STAMPS_US)
    model = sum(stages.values())
    achieved = 1e3 * ms_per_match / super_steps
    return {"bound": "latency", "unit": "us per super-step", "model": model, "achieved": achieved,
            "frac": model / achieved, "super_steps_per_match": super_steps,
            "form": "one co-resident launch per match" if resident else "a kernel per super-step",
            "model_stages_us": stages, "stamped_stages_us": stamps,
            "note": "achieved = median ms per match / mean super-steps per match (includes the host's enqueue and the "
                    "result read-back); stamped = in-kernel wall_clock64 timeline of one scoring workgroup"}


def sweep_ceiling(pkg, ctx, cfg, sc, scan_n, n_poses, launches, bpu, torch):
    """Kernel ceiling beside the matcher-mode number: the same scoring kernel on flat batches of
    device-resident poses (no host round trip, launches back to back)."""
    rs = np.random.RandomState(11)
    poses = torch.from_numpy(sc["init_pose"] + rs.randn(n_poses, 3) * [0.2, 0.2, 0.1]).cuda()
    scores = torch.empty(n_poses, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    for _ in range(5):
        ctx.score_poses_device(0, cfg, n_poses, poses.data_ptr(), scores.data_ptr())
    ctx.synchronize()
    ctx.profile_enable(True)
    ctx.profile_read(reset=True)
    for _ in range(launches):
        ctx.score_poses_device(0, cfg, n_poses, poses.data_ptr(), scores.data_ptr())
    ctx.synchronize()
    ctx.profile_enable(False)
    ms, n, units = ctx.profile_read(reset=True)
    achieved = units * bpu / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    traffic, src = load_traffic("sweep")
    out = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": src,
           "kernel": "k_score_point", "bytes_per_unit": bpu, "launches": n,
           "poses_per_launch": n_poses, "beams": scan_n, "avg_launch_us": 1e3 * ms / max(n, 1)}
    alg = float(n_poses) * scan_n * bpu
    if traffic and traffic < 0.25 * alg:
        out["note"] = ("measured HBM traffic is %.0fx below the algorithmic bytes: the gathers of nearby poses are "
                       "cache hits, the kernel is bound by VALU issue and gather latency (valu)" % (alg / traffic))
    rv = roofline_valu("sweep", out["avg_launch_us"])
    if rv:
        out["valu"] = rv
    return out


N_SCENES = 16


def rotating_scenes(sc, n_beams, weighting, n=N_SCENES):
    """What a robot sees instead of one match repeated: `n` (scan, odometry error) pairs on the scene's map -- robot
    poses jittered around the mapped one, a fresh N(0, 0.01 m) range-noise seed per scan, initial-pose errors from
    zero to three times the default (+0.07 m, -0.04 m, +0.03 rad), in a fixed shuffled order.  Deterministic (the
    CPU baselines' worker processes rebuild the same set)."""
    from synth import cast_scan, viny_weights
    m = sc["map"]
    rs = np.random.RandomState(2024)
    mags = np.linspace(0.0, 3.0, n)
    rs.shuffle(mags)
    out = []
    for j in range(n):
        true = sc["true_pose"] + rs.randn(3) * [0.15, 0.15, 0.04]
        # the scan as the scanner hands it over: every beam, with a flag on the ones that hit something (what
        # TransformedLaserScan holds, sensor_data.h:203-208) -- and the hits alone, i.e. what filter_scan keeps
        raw_rng, raw_ang, occ = cast_scan(sc["gt"], m.scale, true, n_beams, seed=1000 + j, raw=True)
        keep = occ != 0
        rng, ang = raw_rng[keep], raw_ang[keep]
        w = np.full(rng.size, 1.0 / rng.size) if weighting == "even" else viny_weights(rng, ang)
        out.append(dict(range=rng, angle=ang, weight=w, init_pose=true + mags[j] * np.array([0.07, -0.04, 0.03]),
                        raw_range=raw_rng, raw_angle=raw_ang, is_occ=occ, error_x_default=float(mags[j])))
    return out


def physical_cores():
    """(physical cores, logical cores) of this host from /proc/cpuinfo."""
    logical = os.cpu_count() or 1
    try:
        seen, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core = line.split(":", 1)[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        if seen:
            return len(seen), logical
    except OSError:
        pass
    return logical, logical


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


# ------------------------------------------------------------------------------------------- CPU baselines
def _ref_match_worker(job):
    """One worker process: the compiled reference's process_scan over the rotating scenes, for `seconds`.  Map and
    scenes come from a file the parent wrote (rebuilding the synthetic scene would cost every worker half a minute)."""
    path, kind, params, seconds, weighting, first = job
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from synth import MapData
    z = np.load(path)
    m = MapData(int(z["cell_model"]), z["payload"], z["origin"], float(z["scale"]), z["unknown"])
    scenes = [dict(range=z["range%d" % k], angle=z["angle%d" % k], weight=z["weight%d" % k], init_pose=z["init%d" % k],
                   raw_range=z["rrange%d" % k], raw_angle=z["rangle%d" % k], is_occ=z["occ%d" % k])
              for k in range(int(z["n_scenes"]))]
    r = cpu_baseline_reference({"map": m}, kind, params, seconds, weighting, scenes, first)
    return (r["_units"], r["_seconds"]) if r else None


def _save_scene_for_workers(sc, scenes):
    import tempfile
    m = sc["map"]
    d = dict(cell_model=np.array(m.cell_model), payload=m.payload, origin=np.array(m.origin), scale=np.array(m.scale),
             unknown=m.unknown, n_scenes=np.array(len(scenes)))
    for k, s_ in enumerate(scenes):
        d["range%d" % k], d["angle%d" % k], d["weight%d" % k], d["init%d" % k] = (s_["range"], s_["angle"], s_["weight"],
                                                                                 np.asarray(s_["init_pose"]))
        d["rrange%d" % k], d["rangle%d" % k], d["occ%d" % k] = s_["raw_range"], s_["raw_angle"], s_["is_occ"]
    f = tempfile.NamedTemporaryFile(prefix="slamhip_bench_scene_", suffix=".npz", delete=False)
    f.close()
    np.savez(f.name, **d)
    return f.name


def cpu_baseline_reference(sc, kind, params, seconds, weighting, scenes, first=0):
    """The compiled reference itself (oracle/_ref/libslamref.so = the unmodified reference headers
    built in place; travels to the GPU box prebuilt): the synthetic map is rebuilt as a reference
    UnboundedPlainGridMap (pointer-chasing cells, virtual calls) and the reference's own
    HillClimbingScanMatcher / MonteCarloScanMatcher::process_scan is timed on one thread."""
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    m = sc["map"]
    if m.cell_model != 0 or not po.ref_available():
        return None
    R = po.Ref()
    R.lib.ref_map_update_bulk.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double)]
    rm = R.map_create(po.REF_CELL_AFFINE, po.MAP_UNBOUNDED_PLAIN, m.width, m.height, m.scale)
    geo = rm.geometry()
    if geo["origin"] != tuple(m.origin):
        return None
    pay = m.payload[..., 0]
    iy, ix = np.nonzero(pay != m.unknown[0])
    xy = np.ascontiguousarray(np.stack([ix - m.origin[0], iy - m.origin[1]], axis=1), dtype=np.int32)
    vals = np.ascontiguousarray(pay[iy, ix])
    R.lib.ref_map_update_bulk(rm.h, len(vals), xy.ctypes.data_as(C.POINTER(C.c_int)),
                              vals.ctypes.data_as(C.POINTER(C.c_double)))
    # the RAW scans (every beam + its is_occupied flag): the reference filters inside process_scan
    # (pose_enumeration_scan_matcher.h:38), and that is inside the timed calls on both sides
    scans = [R.scan_create(s["raw_range"], s["raw_angle"], s["is_occ"]) for s in scenes]
    spe = R.spe_create(po.OOPE_OBSTACLE, po.OIE_DISCREPANCY, 1 if weighting == "viny" else 0)
    mt = R.matcher_create({"HC": po.SM_HC, "MC": po.SM_MC}[kind], spe, params)
    units, t_used, reps = 0, 0.0, 0
    per_scene = {}
    t_end = time.perf_counter() + seconds
    while True:
        k = (first + reps) % len(scenes)
        t0 = time.perf_counter()
        r = R.process_scan(mt, scans[k], scenes[k]["init_pose"], rm, cap=4)
        t_used += time.perf_counter() - t0
        units += r["n_calls"] * r["filtered_n"]
        if k not in per_scene and kind == "HC":  # (a Monte-Carlo matcher's engine runs on from match to match)
            per_scene[k] = dict(prob=float(r["prob"]), delta=[float(x) for x in r["delta"]], n_calls=int(r["n_calls"]),
                                filtered_n=int(r["filtered_n"]))
        reps += 1
        if (time.perf_counter() > t_end and (kind != "HC" or len(per_scene) == len(scenes))) or reps >= 2000:
            break
    phys, logical = physical_cores()
    return {"value": units / t_used, "unit": "pose-candidates*beams/s", "cores": 1, "kind": "reference",
            "sample": "%d x %s %s process_scan of the compiled reference (oracle/_ref, g++ -O3) over the same %d rotating "
                      "(scan, odometry error) pairs on the same map rebuilt as UnboundedPlainGridMap<AffineQualityMergeCell>, "
                      "%.1f s; host CPU: %s, %d physical / %d logical cores"
                      % (reps, kind, params, len(scenes), t_used, cpu_model(), phys, logical),
            "_units": units, "_seconds": t_used, "_per_scene": per_scene}


def run_workers(fn, jobs):
    """`len(jobs)` worker processes (spawn: fresh interpreters; this process has not touched the GPU yet)."""
    import multiprocessing as mp
    with mp.get_context("spawn").Pool(len(jobs)) as pool:
        return pool.map(fn, jobs)


def cpu_baseline(sc, sc_args, kind, params, seconds, weighting, procs, scenes):
    """Single-thread CPU checker on the same rotating scenes: whole process_scan calls, bounded to ~seconds; plus, for
    context, the same on `procs` host cores at once (independent matches, one per process; 0 = one per PHYSICAL
    core of this host)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    from synth import Scan
    try:
        ref = cpu_baseline_reference(sc, kind, params, seconds, weighting, scenes)
    except Exception as e:  # noqa: BLE001  (a missing/foreign prebuilt .so must not kill the bench)
        print("bench.py: reference baseline unavailable (%s); using the port" % e, file=sys.stderr)
        ref = None
    O = po.Oracle()
    okind = {"HC": po.SM_HC, "MC": po.SM_MC}[kind]
    cfg = po.make_cfg()
    units, t_used, reps = 0, 0.0, 0
    e = O.enumerator(okind, params)
    t_end = time.perf_counter() + (min(seconds, 3.0) if ref is not None else seconds)
    oscans = [Scan(s["range"], s["angle"], s["weight"]) for s in scenes]
    port_scene = {}
    while True:
        k = reps % len(scenes)
        t0 = time.perf_counter()
        r = O.process_scan(e, sc["map"], oscans[k], cfg, scenes[k]["init_pose"], cap=8)
        t_used += time.perf_counter() - t0
        units += r["n_calls"] * oscans[k].n
        if k not in port_scene and kind == "HC":
            port_scene[k] = dict(prob=float(r["prob"]), delta=[float(x) for x in r["delta"]], n_calls=int(r["n_calls"]),
                                 filtered_n=int(oscans[k].n))
        reps += 1
        if (time.perf_counter() > t_end and (ref is not None or kind != "HC" or len(port_scene) == len(scenes))) or reps >= 2000:
            break
    port = {"value": units / t_used, "unit": "pose-candidates*beams/s", "cores": 1, "kind": "port",
            "sample": "%d x process_scan (%s %s) over the same rotating scenes, %.1f s, oracle/slam_oracle.c -O2, "
                      "flat-array map; host CPU: %s, %d logical cores visible"
                      % (reps, kind, params, t_used, cpu_model(), os.cpu_count() or 0)}
    if ref is None:
        port["_per_scene"] = port_scene
        return port
    ref.pop("_units", None)
    ref.pop("_seconds", None)
    ref["port_value"] = port["value"]  # the flat-array C restatement, for context
    # (ref["_per_scene"]: what the reference returned for every benchmarked scene -- main() checks the HIP results of the
    # same scenes against it after the timed region and takes the key out of the line)
    phys, logical = physical_cores()
    procs = phys if procs <= 0 else max(1, min(procs, logical))
    if procs > 1:
        try:
            t0 = time.perf_counter()
            per = min(seconds, 6.0)
            scene_file = _save_scene_for_workers(sc, scenes)
            try:
                res = [x for x in run_workers(_ref_match_worker,
                                              [(scene_file, kind, params, per, weighting, 3 * w) for w in range(procs)]) if x]
            finally:
                os.unlink(scene_file)
            if res:
                ref["all_cores"] = {"value": sum(u / s for u, s in res), "unit": ref["unit"], "cores_used": len(res),
                                    "physical_cores": phys, "logical_cores": logical,
                                    "sample": "%d worker processes (one per physical core unless --cpu-procs says "
                                              "otherwise), each the same reference match loop over the rotating scenes "
                                              "for %.0f s (independent scans: the single-hypothesis matcher has no "
                                              "parallel form); wall %.1f s incl. start-up"
                                              % (len(res), per, time.perf_counter() - t0)}
        except Exception as ex:  # noqa: BLE001
            ref["all_cores"] = {"error": str(ex)}
    return ref


def _ref_pf_worker(job):
    sc_args, n, size, scale, seconds, seed0 = job
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from synth import make_scene
    sc = make_scene(**sc_args)
    return pf_reference_loop(sc, n, size, scale, seconds, seed0)


def pf_reference_loop(sc, n, size, scale, seconds, seed0=1000):
    """(particles x steps, seconds, steps) of the compiled reference's GmappingParticleFilter (shared map, map
    update inside the step -- its default behaviour) on the scan sequence of the PF legs."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    if not po.ref_available():
        return None
    R = po.Ref()
    gp = [0.0, 0.1, 0.0, 0.03, 0.0, 0.0, 0.0, 0.0]
    g = po.RefGmapping(R, n, size, size, scale, gp, np.arange(seed0, seed0 + n, dtype=np.uint32))
    scan = R.scan_create(sc["scan"].range, sc["scan"].angle)
    g.step(scan, sc["true_pose"], 7, np.arange(5000, 5000 + n, dtype=np.uint32))  # builds the map
    rs = np.random.RandomState(5)
    steps, t_used = 0, 0.0
    while t_used < seconds and steps < 40:
        d = rs.randn(3) * [0.05, 0.05, 0.02]
        t0 = time.perf_counter()
        g.step(scan, d, 8 + steps, np.arange(6000 + 100 * steps, 6000 + 100 * steps + n, dtype=np.uint32))
        t_used += time.perf_counter() - t0
        steps += 1
    return n * steps, t_used, steps


def pf_cpu_baselines(args, sc, sc_args, seconds):
    """cfg4 on the host: the compiled reference's filter on the benchmarked map size, single thread (8
    particles are enough: its cost is linear in the particle count, the particles run one after the other),
    and -- for context -- one particle per worker process on `--cpu-procs` cores."""
    try:
        r = pf_reference_loop(sc, 8, args.pf_size, args.scale, seconds)
        if not r:
            return None
        out = {"value": r[0] / r[1], "unit": "particles/s", "cores": 1, "kind": "reference",
               "sample": "%d GmappingParticleFilter steps of 8 particles of the compiled reference (oracle/_ref) on "
                         "the %dx%d map, map update inside the step, %.1f s; host CPU: %s"
                         % (r[2], args.pf_size, args.pf_size, r[1], cpu_model())}
        phys, logical = physical_cores()
        # (each worker builds its own 4000^2 reference map of heap-allocated cells, ~1.3 GB: at most 64 of them)
        procs = min(phys, args.particles, 64) if args.cpu_procs <= 0 else max(1, min(args.cpu_procs, logical, args.particles))
        if procs > 1:
            t0 = time.perf_counter()
            res = [x for x in run_workers(_ref_pf_worker, [(sc_args, 1, args.pf_size, args.scale, min(seconds, 4.0),
                                                            1000 + k) for k in range(procs)]) if x]
            if res:
                out["all_cores"] = {
                    "value": sum(u / s for u, s, _ in res), "unit": "particles/s", "cores_used": len(res),
                    "physical_cores": phys, "logical_cores": logical,
                    "sample": "one particle per worker process (%d processes, each its own reference filter and map: "
                              "the reference itself runs its particles sequentially on one shared map); wall %.1f s "
                              "incl. start-up" % (len(res), time.perf_counter() - t0)}
        return out
    except Exception as e:  # noqa: BLE001
        return {"error": str(e)}


# ------------------------------------------------------------------------------------------- particle filter
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
    if in_library:
        moved_bytes = pfm.migration_stats()["tile_bytes_sent"]
    tt = torch.tensor([dm, float(moved_bytes)], dtype=torch.float64, device=dev)
    mx = tt.clone()
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    sm = tt.clone()
    dist.all_reduce(sm, op=dist.ReduceOp.SUM)
    st = pfm.particle_map_stats()
    out = {"value": n * msteps / mx[0].item(), "unit": "particles/s", "ms_per_step": 1e3 * mx[0].item() / msteps,
           "steps": msteps, "resamplings": resamplings, "map_bytes_moved_between_ranks": sm[1].item(),
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
    rs = np.random.RandomState(8)
    deltas = [sc["true_pose"]] + [rs.randn(3) * [0.03, 0.03, 0.01] for _ in range(args.cfg5_steps + 4)]
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


def k6_roofline(ctx, note, leg=None):
    """roofline object of the map update from the HIP events recorded around every K6 pipeline since the last
    reset (slamhip_profile_read_map_update); `traffic`: HBM bytes per pipeline from the committed PMC passes of
    that leg (all of the pipeline's dispatches added up)."""
    ms, calls, records = ctx.profile_read_map_update(reset=True)
    if not calls or ms <= 0:
        return None
    achieved = records * K6_BYTES_PER_RECORD / (ms * 1e-3) / 1e9
    traffic, traffic_src = load_traffic("k6_" + leg) if leg else (None, None)
    measured = {}
    if traffic:
        # what the memory system really moved per pipeline (PMC passes of the same leg) over the live pipeline time
        gbs = traffic / (ms * 1e-3 / calls) / 1e9
        measured = {"hbm_gbs_measured": gbs, "hbm_utilisation": gbs / HBM_PEAK_GBS,
                    "traffic_over_algorithmic": traffic / (records * K6_BYTES_PER_RECORD / calls)}
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": traffic_src, **measured, "kernel": "K6 pipeline (k_mu_count .. k_mu_apply, sort included)",
            "bytes_per_unit": K6_BYTES_PER_RECORD, "unit_of_work": "(beam, cell) record", "launches": calls,
            "units_launched": records, "avg_launch_us": 1e3 * ms / calls,
            "timing": "HIP events recorded around each K6 pipeline on the context's stream, " + note}


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
    deltas = [sc["true_pose"]] + [rs.randn(3) * [0.05, 0.05, 0.02] for _ in range(args.pf_steps + 6)]
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
        ksteps = 3
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
        except pkg.SlamHipError as e:  # e.g. the pool does not fit: report, do not hide
            out["with_particle_maps"] = {"error": str(e)}
    if world > 1 and args.pf_maps_sharded:
        out["with_particle_maps"] = sharded_particle_maps_leg(args, pkg, ctx, gather, counts, gp, seeds, n, first, count,
                                                              rank, world, scan, deltas, dist, torch, dev)
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


# ------------------------------------------------------------------------------------------------- main
def world_cpu_baseline(sc, kind, params, scenes, weighting, scans=30):
    """The single-hypothesis loop of world_leg on one host core with the COMPILED REFERENCE (oracle/_ref/libslamref.so):
    per scan the reference matcher's process_scan on the reference map, then the reference scan adder's append_scan
    from the matched pose -- the two calls SingleStateHypothesisLaserScanGridWorld::handle_observation makes
    (single_state_hypothesis_laser_scan_grid_world.h:52-65) -- over the same rotating scans.  Falls back to the C
    restatement (kind "port") where the prebuilt reference library is missing."""
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    m0 = sc["map"]
    if po.ref_available() and m0.cell_model == 0:
        R = po.Ref()
        R.lib.ref_map_update_bulk.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double)]
        rm = R.map_create(po.REF_CELL_MEAN, po.MAP_UNBOUNDED_PLAIN, m0.width, m0.height, m0.scale)
        if rm.geometry()["origin"] == tuple(m0.origin):
            pay = m0.payload[..., 0]
            iy, ix = np.nonzero(pay != m0.unknown[0])
            xy = np.ascontiguousarray(np.stack([ix - m0.origin[0], iy - m0.origin[1]], axis=1), dtype=np.int32)
            vals = np.ascontiguousarray(pay[iy, ix])
            R.lib.ref_map_update_bulk(rm.h, len(vals), xy.ctypes.data_as(C.POINTER(C.c_int)),
                                      vals.ctypes.data_as(C.POINTER(C.c_double)))
            rscans = [R.scan_create(s["range"], s["angle"]) for s in scenes]
            spe = R.spe_create(po.OOPE_OBSTACLE, po.OIE_DISCREPANCY, 1 if weighting == "viny" else 0)
            mt = R.matcher_create({"HC": po.SM_HC, "MC": po.SM_MC}[kind], spe, params)
            t0 = time.perf_counter()
            for i in range(scans):
                k = i % len(scenes)
                r = R.process_scan(mt, rscans[k], scenes[k]["init_pose"], rm, cap=4)
                R.append_scan(rm, rscans[k], np.asarray(scenes[k]["init_pose"]) + r["delta"])
            dt = time.perf_counter() - t0
            return {"value": scans / dt, "unit": "scans/s", "cores": 1, "kind": "reference",
                    "sample": "%d scans (reference process_scan + reference append_scan on an "
                              "UnboundedPlainGridMap<MeanProbabilityCell>, oracle/_ref, g++ -O3) over the rotating scenes "
                              "on one core, %.2f s" % (scans, dt)}
    from pyoracle_mapupdate import RULE_MEAN, append_scan_ex
    from synth import Scan
    O = po.Oracle()
    e = O.enumerator({"HC": po.SM_HC, "MC": po.SM_MC}[kind], params)
    m = po.GridMapData(m0.cell_model, m0.payload.copy(), m0.origin, m0.scale, m0.unknown)
    aux = np.zeros(m.payload.shape[:2] + (1,))
    oscans = [Scan(s["range"], s["angle"], s["weight"]) for s in scenes]
    t0 = time.perf_counter()
    for i in range(scans):
        k = i % len(scenes)
        r = O.process_scan(e, m, oscans[k], po.make_cfg(), scenes[k]["init_pose"], cap=8)
        append_scan_ex(O, m, aux, RULE_MEAN, np.asarray(scenes[k]["init_pose"]) + r["delta"], oscans[k].range,
                       oscans[k].angle)
    dt = time.perf_counter() - t0
    return {"value": scans / dt, "unit": "scans/s", "cores": 1, "kind": "port",
            "sample": "%d scans (match + map update) with oracle/slam_oracle.c on one core, %.2f s" % (scans, dt)}


def world_leg(args, pkg, ctx, sc, cfg, kind, params, scenes):
    """One hypothesis, scan after scan, everything through the C-ABI: upload the (filtered) scan, match from the
    odometry pose, append the scan to the map from the matched pose -- SingleStateHypothesisLaserScanGridWorld::
    handle_observation (single_state_hypothesis_laser_scan_grid_world.h:52-65) with the map resident in HBM and its
    update queued behind the match (slamhip_map_set_deferred), as host/slamhip_resident_world.h runs it -- over the
    rotating scans (every scan arrives from the host, as a sensor's would).  Parity of this loop against the
    reference's world: tests/test_gpu_world.py."""
    m0 = sc["map"]
    trig = [pkg.beam_trig(s["angle"]) for s in scenes]
    ctx.map_bind(5, m0.cell_model, m0.width, m0.height, m0.origin, m0.scale, m0.unknown)
    ctx.map_upload_window(5, 0, 0, m0.payload)
    ctx.map_set_auto_grow(5, True)
    m = pkg.Matcher(ctx, kind, cfg, params)
    ctx.map_set_deferred(True)
    it = [0]

    def one_scan():
        k = it[0] % len(scenes)
        it[0] += 1
        s, (cos_a, sin_a) = scenes[k], trig[k]
        ctx.scan_upload(s["range"], cos_a, sin_a, s["weight"], None)
        r = m.process_scan(5, s["init_pose"])
        ctx.map_append_scan(5, pkg.RULE_MEAN, s["init_pose"] + r["delta"], s["range"], cos_a, sin_a)

    for _ in range(len(scenes)):
        one_scan()
    updates = ctx.map_drain()
    n = max(2 * len(scenes), args.steps)
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
            "note": "map resident in HBM (MeanProbabilityCell), the update queued behind the match on the context's "
                    "stream and drained at the end of the timed region; %d rotating (scan, odometry error) pairs, each "
                    "scan uploaded from the host inside its step" % len(scenes)}


def replicas_leg(args, pkg, ctx, cfg, params, scenes, bpu, ks=(1, 2, 4, 8, 16)):
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
        n_calls = max(len(blocks), int(np.ceil(max(args.steps, 32) / K)))
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


def dry_ranks_main(args):
    """One rank of `--dry-ranks N` (see the flag's help).  No GPU is touched: the filter shards are created without a
    context (host-only bookkeeping of the C-ABI: plan_resample / export / import), the scan probabilities are injected,
    and what the library's migrate_and_import does with tile buffers over RCCL is walked here with dummy per-particle
    "maps" over gloo send / recv -- same plan rules (csrc/gmapping.cpp: need[r] = sources rank r draws from other
    ranks, ascending; sends ordered by (destination, source))."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if world != args.dry_ranks:
        print("bench.py: --dry-ranks %d but the launcher started %d rank(s)" % (args.dry_ranks, world), file=sys.stderr)
        sys.exit(2)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    import __graft_entry__ as ge
    pkg = ge.load_package()
    n = args.particles
    counts = [n // world + (1 if r < n % world else 0) for r in range(world)]
    firsts = [sum(counts[:r]) for r in range(world)]
    count, first = counts[rank], firsts[rank]
    bounds = np.cumsum(counts)
    owner = lambda j: int(np.searchsorted(bounds, int(j), side="right"))  # noqa: E731
    gp = [0.0, args.pf_sigma_xy, 0.0, args.pf_sigma_th, 0.0, 0.0, 0.0, 0.0]
    seeds = np.arange(1000, 1000 + n, dtype=np.uint32)

    def gather(a):
        a = np.ascontiguousarray(a)
        per = a.size // max(count, 1)
        padded = np.zeros(max(counts) * per, dtype=a.dtype)
        padded[:a.size] = a.ravel()
        t = torch.from_numpy(padded)
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        return np.concatenate([outs[r].numpy()[:counts[r] * per] for r in range(world)])

    def run(first_, count_, gather_, maps):
        """`steps` filter steps on the shard [first_, first_ + count_); maps: {global particle: dummy map bytes}"""
        pf = pkg.GmappingFilter(None, pkg.gmapping_params(gp8=gp), n, seeds[first_:first_ + count_], first=first_, count=count_)
        log, moved = [], 0
        for step in range(args.pf_steps):
            rs = np.random.RandomState(100 + step)
            probs, poses = rs.rand(n) ** 3 + 1e-3, rs.randn(n, 3)
            _, w, _ = pf.state()
            pf.set(poses=poses[first_:first_ + count_], weights=w * probs[first_:first_ + count_])
            _, raw, _ = pf.state()
            all_raw = gather_(raw)
            wn = all_raw / all_raw.sum()
            need = bool(2.0 / np.sum(wn * wn) < n)
            idx = None
            if need:
                idx = pkg.pf_resample(pkg.pf_normalize(all_raw), 7 + step)
                pf.import_(gather_(pf.export()), idx)
                if count_ == n:  # the unsharded checker: maps follow the indices
                    maps = {i: maps[int(idx[i])] for i in range(n)}
                else:
                    needs = [sorted({int(idx[j]) for j in range(firsts[r], firsts[r] + counts[r]) if owner(idx[j]) != r})
                             for r in range(world)]
                    ops, recv = [], {}
                    for r in range(world):  # my sends by (destination, source), my receives by source
                        if r == rank:
                            continue
                        for src in needs[r]:
                            if owner(src) == rank:
                                t = torch.from_numpy(np.frombuffer(maps[src], dtype=np.uint8).copy())
                                ops.append(dist.P2POp(dist.isend, t, r))
                                moved += t.numel()
                    for src in needs[rank]:
                        recv[src] = torch.empty(64, dtype=torch.uint8)
                        ops.append(dist.P2POp(dist.irecv, recv[src], owner(src)))
                    if ops:
                        for wk in dist.batch_isend_irecv(ops):
                            wk.wait()
                    new = {}
                    for j in range(first_, first_ + count_):
                        src = int(idx[j])
                        new[j] = maps[src] if owner(src) == rank else recv[src].numpy().tobytes()
                    maps = new
            p_, w_, m_ = pf.state()
            log.append((need, idx, p_, w_, m_, dict(maps)))
        pf.close()
        return log, moved

    dummy = lambda j: (np.arange(64, dtype=np.uint8) * 3 + j).astype(np.uint8).tobytes()  # noqa: E731
    dist.barrier()
    t0 = time.perf_counter()
    log, moved = run(first, count, gather, {j: dummy(j) for j in range(first, first + count)})
    dist.barrier()
    dt = time.perf_counter() - t0
    ok = True
    ref_log, _ = run(0, n, lambda a: np.asarray(a), {j: dummy(j) for j in range(n)})
    resamplings = 0
    for (need, idx, p_, w_, m_, maps), (rneed, ridx, rp, rw, rm, rmaps) in zip(log, ref_log):
        resamplings += int(rneed)
        ok &= need == rneed and (not need or np.array_equal(idx, ridx))
        ok &= np.array_equal(p_, rp[first:first + count]) and np.array_equal(w_, rw[first:first + count])
        ok &= np.array_equal(m_, rm[first:first + count])
        ok &= all(maps[j] == rmaps[j] for j in range(first, first + count))
    tt = torch.tensor([dt, float(moved), 0.0 if ok else 1.0], dtype=torch.float64)
    mx = tt.clone()
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    sm = tt.clone()
    dist.all_reduce(sm, op=dist.ReduceOp.SUM)
    if rank == 0:
        print(json.dumps({"dry_run": True, "ranks": world, "particles": n, "shards": counts, "steps": args.pf_steps,
                          "resamplings": resamplings, "dummy_map_bytes_moved": sm[1].item(),
                          "ranks_that_disagree_with_the_unsharded_filter": int(sm[2].item()),
                          "ok": bool(sm[2].item() == 0 and resamplings > 0), "seconds": mx[0].item(),
                          "note": "host-only filter shards over gloo, launched like --gpus N; no GPU touched"}))
    dist.destroy_process_group()
    sys.exit(0 if sm[2].item() == 0 else 1)


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
                per_scene_stats[k] = (st_["scorer_calls"], st_["poses_evaluated"], st_["launches"])
            c_, e_, l_ = per_scene_stats[k]
            evaluated[0] += e_
            plain_calls[0] += c_
            super_steps[0] += l_
            return c_ * kept

        def step_resident():
            k = step_i[0] % len(scenes)
            step_i[0] += 1
            ctx.scan_select(k)
            m.process_scan(0, scenes[k]["init_pose"])
            return account(k, beams_of[k])

        def step_raw():
            k = step_i[0] % len(scenes)
            step_i[0] += 1
            kept = raw_upload[k](scenes[k]["init_pose"])
            m.process_scan(0, scenes[k]["init_pose"])
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
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
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
                pf_out["cpu_baseline"] = pf_cpu_out
        if cfg5_out is not None:
            out["cfg5"] = cfg5_out
        if world_out is not None:
            out["world_loop"] = world_out
        if replicas_out is not None:
            out["replicas"] = replicas_out
        for k_, v_ in extra_legs.items():
            if v_ is not None:
                out[k_] = v_
        print(json.dumps(out))

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
            replicas_out = replicas_leg(args, pkg, ctx, cfg, params, scenes, BYTES_PER_UNIT[bkey])
        except pkg.SlamHipError as e:
            replicas_out = {"error": str(e)}

    if watchdog is not None:
        watchdog.cancel()
    if rank == 0:
        emit_line(pf_out, cfg5_out)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()
    if parity is not None and (parity["traces_equal"] != parity["scenes"] or parity["filtered_counts_equal"] != parity["scenes"]):
        print("bench.py: PARITY FAILURE on the benchmarked scenes %r (the line above carries the details)"
              % parity["scenes_differing"], file=sys.stderr)
        sys.exit(4)


if __name__ == "__main__":
    main()
