"""bench_legs.cpu_baselines -- the CPU side of the line: the compiled reference (oracle/_ref) and the C restatement timed
on this box's host cores, on bounded samples of the benchmarked workloads.  The ONLY part of the benchmark that touches
oracle/; bench.py runs all of it BEFORE the process initialises the GPU (worker processes are spawned)."""
import os
import sys
import time

import numpy as np

from .common import ROOT, cpu_model, physical_cores, run_workers


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


def pf_reference_loop(sc, n, size, scale, seconds, seed0=1000, update=True):
    """(particles x steps, seconds, steps) of the compiled reference's GmappingParticleFilter (shared map) on the scan
    sequence of the PF legs.  update: the map update inside the step (the reference's default behaviour); False: the
    likelihood step alone -- the reference's own parameter slam/mapping/max_range = 0 makes its scan adder return at once
    (grid_map_scan_adders.h:140-142) -- on a map the reference's scan adder built beforehand."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    if not po.ref_available():
        return None
    R = po.Ref()
    gp = [0.0, 0.1, 0.0, 0.03, 0.0, 0.0, 0.0, 0.0]
    seeds = np.arange(seed0, seed0 + n, dtype=np.uint32)
    scan = R.scan_create(sc["scan"].range, sc["scan"].angle)
    if update:
        g = po.RefGmapping(R, n, size, size, scale, gp, seeds)
        g.step(scan, sc["true_pose"], 7, np.arange(5000, 5000 + n, dtype=np.uint32))  # builds the map
    else:
        g = po.RefGmapping(R, n, size, size, scale, gp, seeds, map_max_range=0.0)
        mview = g.map()
        for _k in range(3):
            R.append_scan(mview, scan, sc["true_pose"], quality=1.0, blur=0.0)
        g.step(scan, sc["true_pose"], 7, np.arange(5000, 5000 + n, dtype=np.uint32))  # places the particles
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
        # ... and the likelihood step alone (VERDICT r5 "What's weak" 5: the GPU figure beside it is likelihood-only)
        try:
            rl = pf_reference_loop(sc, 8, args.pf_size, args.scale, min(seconds, 5.0), update=False)
            if rl:
                out["likelihood_only"] = {
                    "value": rl[0] / rl[1], "unit": "particles/s", "cores": 1, "kind": "reference",
                    "sample": "%d steps of 8 particles of the compiled reference with its scan adder switched off through "
                              "its own parameter (slam/mapping/max_range = 0) on the %dx%d map, %.1f s"
                              % (rl[2], args.pf_size, args.pf_size, rl[1])}
        except Exception as ex:  # noqa: BLE001
            out["likelihood_only"] = {"error": str(ex)}
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




VINY_ADDER = dict(quality=0.9, base=(0.95, 0.04, 0.01, 0.003), blur=0.3)  # config/slams/viny_slam_base.properties


def mc_reference_baseline(sc, scenes, params, n_scenes=8):
    """BASELINE configs[2] on one host core with the COMPILED REFERENCE: vinySLAM's map -- an
    UnboundedPlainGridMap<TbmOccConsistentCell> built by the reference's own scan adder (const estimator 0.95 / 0.04,
    0.01 / 0.003, blur 0.3: viny_slam_base.properties) from five of the rotating scans at their true poses -- and the
    reference's MonteCarloScanMatcher::process_scan (viny weights) on the first `n_scenes` raw scans, a FRESH matcher
    per scene (seed params[0]) so that the engine's state is a function of the scene.  Returns the timing, the map as
    a flat payload (the GPU leg uploads THIS map) and what the reference returned per scene (the leg's parity gate)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    if not po.ref_available():
        return None
    R = po.Ref()
    m0 = sc["map"]
    rm = R.map_create(po.REF_CELL_TBM, po.MAP_UNBOUNDED_PLAIN, m0.width, m0.height, m0.scale)
    if rm.geometry()["origin"] != tuple(m0.origin):
        return None
    for s_ in scenes[:5]:
        R.append_scan(rm, R.scan_create(s_["range"], s_["angle"]), s_["true_pose"], VINY_ADDER["quality"], 0,
                      VINY_ADDER["base"], VINY_ADDER["blur"])
    g = rm.geometry()
    if (g["width"], g["height"], g["origin"]) != (m0.width, m0.height, tuple(m0.origin)):
        return None  # (the map grew: the synthetic window no longer describes it)
    md = rm.to_data()
    spe = R.spe_create(po.OOPE_OBSTACLE, po.OIE_DISCREPANCY, 1)
    units, t_used, per_scene = 0, 0.0, {}
    for k, s_ in enumerate(scenes[:n_scenes]):
        mt = R.matcher_create(po.SM_MC, spe, params)
        rscan = R.scan_create(s_["raw_range"], s_["raw_angle"], s_["is_occ"])
        t0 = time.perf_counter()
        r = R.process_scan(mt, rscan, s_["init_pose"], rm, cap=4)
        t_used += time.perf_counter() - t0
        units += r["n_calls"] * r["filtered_n"]
        per_scene[k] = dict(prob=float(r["prob"]), delta=[float(x) for x in r["delta"]], n_calls=int(r["n_calls"]),
                            filtered_n=int(r["filtered_n"]))
    return {"value": units / t_used, "unit": "pose-candidates*beams/s", "cores": 1, "kind": "reference",
            "sample": "%d x MC %s process_scan of the compiled reference (oracle/_ref, g++ -O3), a fresh matcher per scene, on "
                      "an UnboundedPlainGridMap<TbmOccConsistentCell> built by the reference's scan adder from five of the "
                      "scans; viny weights; %.1f s; host CPU: %s" % (len(per_scene), params, t_used, cpu_model()),
            "_map": dict(payload=md.payload, origin=tuple(md.origin), scale=md.scale, unknown=np.asarray(md.unknown)),
            "_per_scene": per_scene}
