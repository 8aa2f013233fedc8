"""GPU suite: K independent hill-climbing matches in shared launches (slamhip_matcher_process_scan_batch,
csrc/hc_chain.hip with grid.y = match and a job table) -- PoseEnumerationScanMatcher::process_scan
(pose_enumeration_scan_matcher.h:31-77) once per robot.  The bar: every match of a batch has the observer trace
(poses, scores, accepted flags), result and scorer-call count of a LONE slamhip_matcher_process_scan on the same
scan / pose / map, bit for bit, whatever the batch size (the speculation tree of a chain shrinks with K, the accept
path may not notice), and the oracle's strict accept loop gives the same trace."""
import ctypes as C

import numpy as np
import pytest
from helpers import assert_trace_equal
from synth import CELL_OCC, CELL_TBM, cast_scan, make_scene, viny_weights

import __graft_entry__ as ge

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    return ge.load_package()


@pytest.fixture(scope="module")
def ctx(pkg):
    c = pkg.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def tctx(pkg):
    """a context of libslamhip_testing.so -- the build with the test hooks (slamhip_matcher_debug_*) compiled in"""
    c = pkg.Context(0, testing=True)
    yield c
    c.close()


def scenes(pkg, ctx, cell, weighting, k, beams=(720,), sizes=(600, 500)):
    """k jobs over len(sizes) maps: scans cast from jittered robot poses with their own noise seeds, initial poses
    with errors from zero to three times the default."""
    worlds = []
    for mi, size in enumerate(sizes):
        sc = make_scene(cell_model=cell, size=size, scale=0.05, n_beams=max(beams), seed=5 + mi, weighting=weighting)
        ctx.upload_map(mi, sc["map"])
        worlds.append(sc)
    rs = np.random.RandomState(100 + k)
    jobs = []
    for j in range(k):
        sc = worlds[j % len(worlds)]
        true = sc["true_pose"] + rs.randn(3) * [0.15, 0.15, 0.05]
        nb = beams[j % len(beams)]
        rng, ang = cast_scan(sc["gt"], 0.05, true, nb, seed=42 + j)
        w = np.full(rng.size, 1.0 / rng.size) if weighting == "even" else viny_weights(rng, ang)
        cos_a, sin_a = pkg.beam_trig(ang)
        err = np.array([0.07, -0.04, 0.03]) * (3.0 * j / max(k - 1, 1))
        jobs.append(dict(map_id=j % len(worlds), range=rng, cos_a=cos_a, sin_a=sin_a, weight=w, factor=None,
                         init_pose=true + err, angle=ang, world=sc))
    return jobs


def lone(pkg, ctx, m, job, trace=True):
    ctx.scan_upload(job["range"], job["cos_a"], job["sin_a"], job["weight"], np.ones(job["range"].size))
    return m.process_scan(job["map_id"], job["init_pose"], trace=trace)


@pytest.mark.parametrize("mode", [1, 2])  # kernel chains (hc_chain.hip) / one co-resident launch (hc_resident.hip)
@pytest.mark.parametrize("cell,weighting", [(CELL_OCC, "even"), (CELL_TBM, "viny")])
@pytest.mark.parametrize("k", [2, 3, 8, 20])
def test_batch_equals_lone_matches(pkg, ctx, cell, weighting, k, mode):
    jobs = scenes(pkg, ctx, cell, weighting, k, beams=(720, 360, 1080))
    prm = [24, 0.1, 0.1]
    mb = pkg.Matcher(ctx, "HC", pkg.spe_cfg(), prm)
    mb.set_device_chain(mode)
    ml = pkg.Matcher(ctx, "HC", pkg.spe_cfg(), prm)
    ml.set_device_chain(1)
    got = mb.process_scan_batch(jobs, trace=True)
    assert len(got) == k
    total_calls = 0
    for j, (g, job) in enumerate(zip(got, jobs)):
        want = lone(pkg, ctx, ml, job)
        assert_trace_equal(g, want)  # bit for bit
        st = mb.batch_stats(j)
        assert st["on_device_chain"] and st["scorer_calls"] == want["n_calls"] == ml.stats()["scorer_calls"]
        total_calls += st["scorer_calls"]
    s = mb.stats()
    assert s["scorer_calls"] == total_calls
    if mode == 2:  # all chains' workgroups fit the device at once: ONE launch, nobody gave up
        assert s["kernels_launched"] == 1 and mb.resident_stats() == dict(matches=1, gave_up=0)
    else:
        assert s["kernels_launched"] >= 3
    # without an observer: same results, no trace buffers involved; the argument block reused
    blk = mb.make_batch(jobs)
    for _ in range(2):
        again = mb.process_scan_batch(blk)
        for g, a in zip(got, again):
            assert g["prob"] == a["prob"] and np.array_equal(g["delta"], a["delta"])


def test_batch_against_the_oracle(pkg, ctx, oracle):
    """The oracle's strict accept loop (beam-order sums, libm trigonometry) on every job of a batch: identical
    trace, scores to 1e-12 -- the default mode's contract, through the batch entry point."""
    import pyoracle as po
    from synth import Scan
    jobs = scenes(pkg, ctx, CELL_OCC, "even", 6)
    prm = [16, 0.1, 0.1]
    mb = pkg.Matcher(ctx, "HC", pkg.spe_cfg(), prm)
    got = mb.process_scan_batch(jobs, trace=True)
    for g, job in zip(got, jobs):
        e = oracle.enumerator(po.SM_HC, prm)
        scan = Scan(job["range"], job["angle"], job["weight"])
        r = oracle.process_scan(e, job["world"]["map"], scan, po.make_cfg(), job["init_pose"])
        assert_trace_equal(g, r, exact_scores=False, rtol=1e-12)


def test_batch_falls_back_job_by_job(pkg, tctx):
    """What the shared launches do not cover runs through the single-match path inside the same call: a strict-mode
    matcher (beam-order sum, host trigonometry) keeps every job off the chains; a trace buffer made too small
    (testing hook) sends the jobs whose trace outgrew it there and leaves the others on the chains."""
    ctx = tctx
    jobs = scenes(pkg, ctx, CELL_OCC, "even", 4)
    prm = [12, 0.1, 0.1]
    strict = dict(sum_order=1, pose_trig=1)
    mb = pkg.Matcher(ctx, "HC", pkg.spe_cfg(**strict), prm)
    ml = pkg.Matcher(ctx, "HC", pkg.spe_cfg(**strict), prm)
    got = mb.process_scan_batch(jobs, trace=True)
    for j, (g, job) in enumerate(zip(got, jobs)):
        assert_trace_equal(g, lone(pkg, ctx, ml, job))
        assert not mb.batch_stats(j)["on_device_chain"]
    md = pkg.Matcher(ctx, "HC", pkg.spe_cfg(), prm)
    ml = pkg.Matcher(ctx, "HC", pkg.spe_cfg(), prm)
    want = [lone(pkg, ctx, ml, job) for job in jobs]
    cap = sorted(w["n_calls"] for w in want)[1] + 1  # the two shortest traces fit
    L = pkg.load(testing=True)
    L.slamhip_matcher_debug_trace_cap.argtypes = [C.c_void_p, C.c_int]
    assert L.slamhip_matcher_debug_trace_cap(md.h, cap) == 0
    got = md.process_scan_batch(jobs, trace=True)
    on = [md.batch_stats(j)["on_device_chain"] for j in range(4)]
    assert 1 <= sum(on) < 4, on
    for g, w in zip(got, want):
        assert_trace_equal(g, w)


def test_batch_argument_checks(pkg, ctx):
    jobs = scenes(pkg, ctx, CELL_OCC, "even", 2)
    m = pkg.Matcher(ctx, "HC", pkg.spe_cfg(), [6, 0.1, 0.1])
    bad = [dict(j) for j in jobs]
    bad[1]["map_id"] = 17
    with pytest.raises(pkg.SlamHipError):
        m.process_scan_batch(bad)
    assert m.process_scan_batch([]) == []
    one = m.process_scan_batch(jobs[:1], trace=True)  # a batch of one is a lone match
    assert_trace_equal(one[0], lone(pkg, ctx, pkg.Matcher(ctx, "HC", pkg.spe_cfg(), [6, 0.1, 0.1]), jobs[0]))


def test_stored_scans_equal_uploaded_scans(pkg, ctx):
    """slamhip_scan_store / slamhip_scan_select: a scan kept in HBM and selected gives the lone match of the same
    scan uploaded, and a batch whose jobs name scan slots equals the batch whose jobs carry the arrays."""
    jobs = scenes(pkg, ctx, CELL_OCC, "even", 5, beams=(720, 1080))
    prm = [16, 0.1, 0.1]
    ml = pkg.Matcher(ctx, "HC", pkg.spe_cfg(), prm)
    want = [lone(pkg, ctx, ml, job) for job in jobs]
    for k, job in enumerate(jobs):
        ctx.scan_store(10 + k, job["range"], job["cos_a"], job["sin_a"], job["weight"])
    for k in (3, 0, 4):
        ctx.scan_select(10 + k)
        assert_trace_equal(ml.process_scan(jobs[k]["map_id"], jobs[k]["init_pose"], trace=True), want[k])
    mb = pkg.Matcher(ctx, "HC", pkg.spe_cfg(), prm)
    slot_jobs = [dict(map_id=j["map_id"], scan_slot=10 + k, init_pose=j["init_pose"]) for k, j in enumerate(jobs)]
    for g, w in zip(mb.process_scan_batch(slot_jobs, trace=True), want):
        assert_trace_equal(g, w)
    mixed = [slot_jobs[0], jobs[1], slot_jobs[2], jobs[3], slot_jobs[4]]
    for g, w in zip(mb.process_scan_batch(mixed, trace=True), want):
        assert_trace_equal(g, w)
    # a stored scan rewritten in place while selected; an empty slot is an error
    ctx.scan_select(10)
    ctx.scan_store(10, jobs[1]["range"], jobs[1]["cos_a"], jobs[1]["sin_a"], jobs[1]["weight"])
    assert_trace_equal(ml.process_scan(jobs[1]["map_id"], jobs[1]["init_pose"], trace=True), want[1])
    with pytest.raises(pkg.SlamHipError):
        ctx.scan_select(999)
    with pytest.raises(pkg.SlamHipError):
        mb.process_scan_batch([dict(map_id=0, scan_slot=998, init_pose=[0, 0, 0])])


def test_resident_batch_gives_up_when_a_workgroup_is_missing(pkg, tctx):
    """The co-resident batch launch (csrc/hc_resident.hip) with one workgroup of every chain leaving at once (testing
    hook): the chains must give up within the bound, and the kernel chains redo the whole batch with the lone
    matches' traces."""
    ctx = tctx
    import time
    jobs = scenes(pkg, ctx, CELL_OCC, "even", 4)
    prm = [16, 0.1, 0.1]
    mb = pkg.Matcher(ctx, "HC", pkg.spe_cfg(), prm)
    mb.set_device_chain(2)
    ml = pkg.Matcher(ctx, "HC", pkg.spe_cfg(), prm)
    L = pkg.load(testing=True)
    L.slamhip_matcher_debug_resident_mute.argtypes = [C.c_void_p, C.c_int]
    L.slamhip_matcher_debug_resident_mute.restype = C.c_int
    assert L.slamhip_matcher_debug_resident_mute(mb.h, 3) == 0
    t0 = time.time()
    got = mb.process_scan_batch(jobs, trace=True)
    assert time.time() - t0 < 5.0
    assert mb.resident_stats() == dict(matches=1, gave_up=1) and mb.stats()["kernels_launched"] >= 3
    for g, job in zip(got, jobs):
        assert_trace_equal(g, lone(pkg, ctx, ml, job))
    assert L.slamhip_matcher_debug_resident_mute(mb.h, 0) == 0
    again = mb.process_scan_batch(jobs, trace=True)
    for g, a in zip(got, again):
        assert_trace_equal(g, a)
    assert mb.resident_stats() == dict(matches=2, gave_up=1) and mb.stats()["kernels_launched"] == 1


@pytest.mark.parametrize("k", [2, 8])
def test_batch_inert_tails_equal_lone_matches_with_every_call_scored(pkg, ctx, k):
    """r06: the chains of a batch end on their inert roots too (csrc/hc_resident.hip: the closed-form tail of a
    hill-climbing match, SLAMHIP_OPT_INERT_TAIL) -- each chain's bookkeeping workgroup writes its tail into its own
    stretch of the observers' trace buffer.  k matches at limit 128 (k = 8: the pair form, two poses per workgroup) against
    lone matches on the kernel chain with every call scored (option 0): traces bit for bit, counts equal, and the batch's
    closed-form calls are what its chains did not score."""
    jobs = scenes(pkg, ctx, CELL_OCC, "even", k, beams=(720, 360, 1080))
    prm = [128, 0.1, 0.1]
    mb = pkg.Matcher(ctx, "HC", pkg.spe_cfg(), prm)
    mb.set_device_chain(2)
    ml = pkg.Matcher(ctx, "HC", pkg.spe_cfg(), prm)
    ml.set_device_chain(1)
    got = mb.process_scan_batch(jobs, trace=True)
    sb = mb.stats()
    assert mb.resident_stats() == dict(matches=1, gave_up=0)
    evaluated = 0
    for j, (g, job) in enumerate(zip(got, jobs)):
        ctx.set_option(pkg.OPT_INERT_TAIL, 0)  # (the lone matches: every call scored)
        try:
            want = lone(pkg, ctx, ml, job)
        finally:
            ctx.set_option(pkg.OPT_INERT_TAIL, 2)
        assert_trace_equal(g, want)
        st = mb.batch_stats(j)
        assert st["scorer_calls"] == want["n_calls"] == ml.stats()["scorer_calls"]
        assert ml.stats()["calls_closed_form"] == 0
        evaluated += st["poses_evaluated"]
    assert sb["calls_closed_form"] >= k * 6 * 20 and sb["calls_closed_form"] < sb["scorer_calls"]
    mb.close()
    ml.close()


@pytest.mark.parametrize("cell,weighting", [(CELL_OCC, "even"), (CELL_TBM, "viny")])
def test_batch_certified_tails_fuzz_against_every_call_scored(pkg, ctx, cell, weighting):
    """r06: a batch's chains also end on CERTIFIED poses (SLAMHIP_OPT_INERT_TAIL 2, the default; csrc/hc_resident.hip
    "certificate"): a workgroup says beside its score from which failed-round count on no candidate of a round based on
    its pose can leave the pose's cells -- an argument about rounding, so: batches of 2 ... 32 matches over limits, step
    sizes and beam counts, each run at levels 2, 1 (identical poses only) and 0 (every call scored); traces, results and
    counts assert-equal, and level 2 really ends chains earlier than level 1."""
    rs = np.random.RandomState(5)
    closed = {1: 0, 2: 0}
    steps = {0: 0, 1: 0, 2: 0}
    for it, k in enumerate([2, 3, 8, 16, 32, 5, 12, 24]):
        jobs = scenes(pkg, ctx, cell, weighting, k, beams=(720, 360, 1080))
        prm = [int(rs.choice([70, 128, 200])), float(rs.choice([0.2, 0.1, 0.02])), float(rs.choice([0.1, 0.05, 0.005]))]
        res = {}
        for level in (2, 1, 0):
            ctx.set_option(pkg.OPT_INERT_TAIL, level)
            try:
                mb = pkg.Matcher(ctx, "HC", pkg.spe_cfg(), prm)
                mb.set_device_chain(2)
                got = mb.process_scan_batch(jobs, trace=True)
                st = [mb.batch_stats(j) for j in range(k)]
                res[level] = (got, st, mb.stats())
                # (k = 24 included: r05's tree sizing left batches of 10 and 24 ... 28 matches a few workgroups over what
                # is resident together, and they ran as kernel chains)
                assert mb.resident_stats() == dict(matches=1, gave_up=0), (k, level, prm)
                quiet = mb.process_scan_batch(jobs)
                for g, q in zip(got, quiet):
                    assert g["prob"] == q["prob"] and np.array_equal(g["delta"], q["delta"])
                mb.close()
            finally:
                ctx.set_option(pkg.OPT_INERT_TAIL, 2)
        for lv in (2, 1):
            for (a, sa), (b, sb) in zip(zip(res[lv][0], res[lv][1]), zip(res[0][0], res[0][1])):
                assert_trace_equal(a, b)
                assert a["prob"] == b["prob"] and np.array_equal(a["delta"], b["delta"]) and a["n_calls"] == b["n_calls"]
                assert sa["scorer_calls"] == sb["scorer_calls"]
            closed[lv] += res[lv][2]["calls_closed_form"]
        for lv in (0, 1, 2):
            steps[lv] += res[lv][2]["launches"]
        assert res[0][2]["calls_closed_form"] == 0
    assert closed[2] > closed[1] > 0 and steps[2] < steps[1] < steps[0]
