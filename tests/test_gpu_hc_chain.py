"""GPU suite: hill climbing kept on the device (csrc/hc_chain.hip) against the host-driven matcher and the
oracle.  The chain kernel scores with the arithmetic of k_score_point and replays the reference's accept
loop (pose_enumeration_scan_matcher.h:31-77) itself, so
  * its observer trace (poses, scores, accepted flags), result pose and score must equal the host-driven
    default mode BIT FOR BIT (same sincos, same canonical sum, same enumerator arithmetic);
  * against the oracle's strict accept loop the trace must be identical with scores within 1e-12;
  * a fuzz over random scenes counts accept-trace divergences between the default mode (device chain) and
    the strict mode (SLAMHIP_SUM_SEQUENTIAL + host trig, bit-exact with the reference): none allowed."""
import os

import numpy as np
import pytest
from helpers import assert_trace_equal
from synth import CELL_OCC, CELL_TBM, make_scene

import __graft_entry__ as ge

pytestmark = pytest.mark.gpu
STRICT = dict(sum_order=1, pose_trig=1)


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


@pytest.fixture(scope="module")
def po():
    import pyoracle
    return pyoracle


def upload(pkg, ctx, sc):
    ctx.upload_map(0, sc["map"])
    cos_a, sin_a = pkg.beam_trig(sc["scan"].angle)
    ctx.scan_upload(sc["scan"].range, cos_a, sin_a, sc["scan"].weight, sc["scan"].factor)


CHAIN_MODES = [1, 2]  # a kernel per super-step (hc_chain.hip) / one co-resident launch (hc_resident.hip)


def matchers(pkg, ctx, prm, threads=0, mode=1):
    dev = pkg.Matcher(ctx, "HC", pkg.spe_cfg(), prm)
    dev.set_device_chain(mode, threads)
    host = pkg.Matcher(ctx, "HC", pkg.spe_cfg(), prm)
    host.set_device_chain(0)
    for m in (dev, host):
        m.set_tie_check(0)  # both decide from the tree sums as they are: bit for bit the same walk
    return dev, host


@pytest.mark.parametrize("mode", CHAIN_MODES)
@pytest.mark.parametrize("cell,weighting", [(CELL_OCC, "even"), (CELL_TBM, "viny")])
@pytest.mark.parametrize("prm", [[1, 0.1, 0.1], [6, 0.1, 0.1], [128, 0.1, 0.1], [250, 0.3, 0.05], [700, 0.1, 0.1],
                                 [1000, 0.2, 0.1]])
def test_chain_equals_host_driven_matcher(pkg, ctx, po, oracle, cell, weighting, prm, mode):
    sc = make_scene(cell_model=cell, size=600, scale=0.05, n_beams=720, seed=5, weighting=weighting)
    upload(pkg, ctx, sc)
    dev, host = matchers(pkg, ctx, prm, mode=mode)
    init = sc["init_pose"]
    for rep in range(3):  # repeated matches on one matcher: run-ahead kernels of the last chain, new epoch
        td = dev.process_scan(0, init, trace=True)
        th = host.process_scan(0, init, trace=True)
        assert_trace_equal(td, th)  # bit for bit
        sd, sh = dev.stats(), host.stats()
        assert sd["scorer_calls"] == sh["scorer_calls"] == td["n_calls"]
        assert sd["launches"] >= 1
        # without an observer: same result, no trace buffer involved
        q = dev.process_scan(0, init)
        assert q["prob"] == td["prob"] and np.array_equal(q["delta"], td["delta"])
        init = init + np.array([0.013, -0.007, 0.004])
    if mode == 2:  # every match ran as ONE launch, none gave up
        assert dev.resident_stats() == dict(matches=6, gave_up=0) and dev.stats()["kernels_launched"] == 1
    # the oracle's strict loop: identical trace, scores to 1e-12
    e = oracle.enumerator(po.SM_HC, prm)
    r = oracle.process_scan(e, sc["map"], sc["scan"], po.make_cfg(), sc["init_pose"])
    t = dev.process_scan(0, sc["init_pose"], trace=True)
    assert_trace_equal(t, r, exact_scores=False, rtol=1e-12)


@pytest.mark.parametrize("mode", CHAIN_MODES)
@pytest.mark.parametrize("threads", [256, 512, 1024])
def test_chain_workgroup_sizes_and_beam_counts(pkg, ctx, threads, mode):
    rs = np.random.RandomState(3)
    for n_beams in (1, 63, 257, 1080, 1500, 2300):
        sc = make_scene(cell_model=CELL_OCC, size=400, scale=0.1, n_beams=max(n_beams, 8), seed=7)
        s = sc["scan"]
        keep = np.sort(rs.choice(s.n, min(n_beams, s.n), replace=False))
        s.range, s.angle, s.weight, s.factor = s.range[keep], s.angle[keep], s.weight[keep], s.factor[keep]
        upload(pkg, ctx, sc)
        dev, host = matchers(pkg, ctx, [20, 0.1, 0.1], threads, mode)
        assert_trace_equal(dev.process_scan(0, sc["init_pose"], trace=True),
                           host.process_scan(0, sc["init_pose"], trace=True))
        # the checked default mode (what a matcher does unless told otherwise) walks the beam-order sum's accept path
        chk = pkg.Matcher(ctx, "HC", pkg.spe_cfg(), [20, 0.1, 0.1])
        chk.set_device_chain(mode, threads)
        seq = pkg.Matcher(ctx, "HC", pkg.spe_cfg(sum_order=1), [20, 0.1, 0.1])
        assert_trace_equal(chk.process_scan(0, sc["init_pose"], trace=True),
                           seq.process_scan(0, sc["init_pose"], trace=True), exact_scores=False, rtol=1e-12)


def test_scans_too_long_for_the_chain_take_the_host_driven_path(pkg, ctx, po, oracle):
    """A pose's beam terms sit in LDS in the chain kernels: beyond 4096 beams a matcher goes through the host-driven
    batches by itself -- same trace as the oracle's loop."""
    sc = make_scene(cell_model=CELL_OCC, size=600, scale=0.05, n_beams=6000, seed=3)
    upload(pkg, ctx, sc)
    for kind, okind, prm in (("HC", po.SM_HC, [8, 0.1, 0.1]), ("MC", po.SM_MC, [17, 0.2, 0.1, 20, 100])):
        m = pkg.Matcher(ctx, kind, pkg.spe_cfg(), prm)
        t = m.process_scan(0, sc["init_pose"], trace=True)
        assert m.stats()["kernels_launched"] == 0  # no chain kernel
        e = oracle.enumerator(okind, prm)
        r = oracle.process_scan(e, sc["map"], sc["scan"], po.make_cfg(), sc["init_pose"])
        assert_trace_equal(t, r, exact_scores=False, rtol=1e-12)


@pytest.mark.parametrize("mode", CHAIN_MODES)
def test_chain_zero_weight_scan_and_far_pose(pkg, ctx, mode):
    sc = make_scene(cell_model=CELL_OCC, size=400, scale=0.1, n_beams=360, seed=9)
    sc["scan"].weight[:] = 0.0  # total weight 0: every score is NaN, nothing is ever accepted
    upload(pkg, ctx, sc)
    dev, host = matchers(pkg, ctx, [6, 0.1, 0.1], mode=mode)
    td, th = dev.process_scan(0, sc["init_pose"], trace=True), host.process_scan(0, sc["init_pose"], trace=True)
    assert td["n_calls"] == th["n_calls"] == 1 + 6 * 6 + 1 and not td["accepted"][1:].any()
    assert np.isnan(td["scores"]).all() and np.array_equal(td["poses"], th["poses"])
    sc = make_scene(cell_model=CELL_OCC, size=400, scale=0.1, n_beams=360, seed=9)
    upload(pkg, ctx, sc)
    far = np.array([1e4, -1e4, 0.3])  # every end point outside the window: unknown cells, all scores equal
    assert_trace_equal(dev.process_scan(0, far, trace=True), host.process_scan(0, far, trace=True))


def test_fuzz_default_mode_takes_the_strict_modes_accept_path(pkg, ctx):
    """does a last-ulp difference of the default mode flip a strict `best < candidate` of the
    bit-exact strict mode (beam-order sum + host trig)?  200 HC and 200 MC matches over 40 random scenes.
      * default mode, CHECKED (what a matcher does unless told otherwise: canonical tree sum, device sincos,
        comparisons the tree sum cannot settle decided again from beam-order sums) -- HC on the device chain, HC
        and MC through host-driven batches: no divergence, and the check did fire;
      * the beam-order sum throughout on the device chain (sum_order = SEQUENTIAL, device sincos): none either --
        the device sincos is not what flips a comparison;
      * for the record, the unchecked default mode (tree sums as they are): the few divergences all sit at
        comparisons whose strict-mode sums are equal or one ulp apart (mathematically tied candidates: the same
        multiset of beam terms met in another beam order)."""
    names = ("hc_dev", "hc_k1", "hc_host", "hc_seq", "hc_raw", "mc", "mc_raw")
    div = dict.fromkeys(names, 0)
    rescored = dict.fromkeys(names, 0)
    matches, calls = 0, 0
    n_scenes = int(os.environ.get("SLAMHIP_FUZZ_SCENES", "40"))  # (a longer soak: 400 scenes, run by hand)
    for seed in range(n_scenes):
        cell = CELL_TBM if seed % 3 == 0 else CELL_OCC
        sc = make_scene(cell_model=cell, size=500, scale=0.05, n_beams=360 + 90 * (seed % 5), seed=100 + seed,
                        weighting="viny" if cell == CELL_TBM else "even")
        upload(pkg, ctx, sc)
        rs = np.random.RandomState(seed)
        prm = [6 + 7 * (seed % 4), 0.1, 0.1]
        ms = dict(hc_dev=pkg.Matcher(ctx, "HC", pkg.spe_cfg(), prm), hc_host=pkg.Matcher(ctx, "HC", pkg.spe_cfg(), prm),
                  hc_seq=pkg.Matcher(ctx, "HC", pkg.spe_cfg(sum_order=1), prm), hc_raw=pkg.Matcher(ctx, "HC", pkg.spe_cfg(), prm),
                  hc_k1=pkg.Matcher(ctx, "HC", pkg.spe_cfg(), prm))
        ms["hc_k1"].set_device_chain(1)  # (hc_dev: the default, one co-resident launch per match)
        ms["hc_host"].set_device_chain(0)
        ms["hc_raw"].set_device_chain(0)
        ms["hc_raw"].set_tie_check(0)
        strict = dict(hc=pkg.Matcher(ctx, "HC", pkg.spe_cfg(**STRICT), prm))
        for rep in range(5):
            # Monte Carlo: fresh matchers per match, so that every one of them draws the same random stream
            mc_prm = [1000 + 5 * seed + rep, 0.2, 0.1, 20 + 10 * (seed % 3), 300]
            ms["mc"], ms["mc_raw"] = pkg.Matcher(ctx, "MC", pkg.spe_cfg(), mc_prm), pkg.Matcher(ctx, "MC", pkg.spe_cfg(), mc_prm)
            ms["mc_raw"].set_tie_check(0)
            strict["mc"] = pkg.Matcher(ctx, "MC", pkg.spe_cfg(**STRICT), mc_prm)
            init = sc["true_pose"] + rs.randn(3) * [0.08, 0.08, 0.04]
            ref = {k: m.process_scan(0, init, trace=True) for k, m in strict.items()}
            matches += 1
            calls += ref["hc"]["n_calls"] + ref["mc"]["n_calls"]
            for which, m in ms.items():
                b = ref[which[:2]]
                a = m.process_scan(0, init, trace=True)
                rescored[which] += m.stats()["steps_rescored"]
                n = min(a["n_calls"], b["n_calls"])
                bad = np.nonzero((a["accepted"][:n] != b["accepted"][:n]) | (a["poses"][:n] != b["poses"][:n]).any(1))[0]
                if a["n_calls"] == b["n_calls"] and len(bad) == 0:
                    np.testing.assert_allclose(a["scores"], b["scores"], rtol=1e-12, atol=0)
                    continue
                div[which] += 1
                if which.endswith("_raw"):
                    i = int(bad[0]) if len(bad) else n
                    assert i < n and np.array_equal(a["poses"][i], b["poses"][i])  # same candidate, other decision
                    acc = np.nonzero(b["accepted"][:i])[0]
                    best = b["scores"][acc[-1]]
                    assert abs(b["scores"][i] - best) <= 2 * np.spacing(best), \
                        "default mode flipped a comparison that is not a tie: %r vs %r" % (b["scores"][i], best)
    print("fuzz: %d matches per matcher, divergences %r, re-scored steps / batches %r" % (matches, div, rescored))
    assert matches == 5 * n_scenes and calls > matches * 80
    for which in ("hc_dev", "hc_k1", "hc_host", "mc"):
        assert div[which] == 0, "%d of %d checked default-mode %s matches diverged from the strict mode" % (div[which], matches, which)
        # (Monte Carlo candidates are continuous random poses: sums that close with different terms are rare)
        assert which == "mc" or rescored[which] > 0, "the scenes never exercised the check of %s" % which
    assert div["hc_seq"] == 0, "%d of %d matches diverged with the beam-order sum on the device" % (div["hc_seq"], matches)
    assert rescored["hc_raw"] == rescored["mc_raw"] == 0
    assert div["hc_raw"] <= matches // 20 and div["mc_raw"] <= matches // 20


@pytest.mark.parametrize("oope", ["max", "mean", "overlap"])
def test_fuzz_default_mode_over_the_window_oopes(pkg, ctx, oope):
    """the reference's strict `best < candidate` (pose_enumeration_scan_matcher.h:56) over the
    window OOPEs (occupancy_observation_probability.h:29-99).  `max` is as discrete as the 1-cell value, so
    mathematically tied candidates -- the same multiset of beam terms met in another beam order -- are as likely.
    200 hill-climbing matches per OOPE over 40 random scenes: the default mode (canonical tree sum, device sincos,
    CHECKED since r05: term-vector fingerprints from K2 and from the WIN instantiation of the co-resident chain) on
    the co-resident launch and through host-driven batches against the strict mode (beam-order sum + host trig =
    the reference's arithmetic): no divergence allowed; the unchecked default mode is counted for the record, and
    whatever it flips must be a tie of the strict sums."""
    kinds = dict(max=pkg.OOPE_MAX, mean=pkg.OOPE_MEAN, overlap=pkg.OOPE_OVERLAP)
    area = (-0.06, 0.06, -0.04, 0.04)
    names = ("dev", "host", "raw")
    div, rescored = dict.fromkeys(names, 0), dict.fromkeys(names, 0)
    matches = calls = 0
    n_scenes = int(os.environ.get("SLAMHIP_FUZZ_SCENES", "40"))
    for seed in range(n_scenes):
        cell = CELL_TBM if seed % 3 == 0 else CELL_OCC
        sc = make_scene(cell_model=cell, size=500, scale=0.05, n_beams=360 + 90 * (seed % 5), seed=300 + seed,
                        weighting="viny" if cell == CELL_TBM else "even")
        upload(pkg, ctx, sc)
        rs = np.random.RandomState(1000 + seed)
        prm = [6 + 7 * (seed % 4), 0.1, 0.1]
        cfg = pkg.spe_cfg(oope=kinds[oope], area=area)
        ms = dict(dev=pkg.Matcher(ctx, "HC", cfg, prm), host=pkg.Matcher(ctx, "HC", cfg, prm), raw=pkg.Matcher(ctx, "HC", cfg, prm))
        ms["host"].set_device_chain(0)
        ms["raw"].set_tie_check(0)
        strict = pkg.Matcher(ctx, "HC", pkg.spe_cfg(oope=kinds[oope], area=area, **STRICT), prm)
        for rep in range(5):
            init = sc["true_pose"] + rs.randn(3) * [0.08, 0.08, 0.04]
            b = strict.process_scan(0, init, trace=True)
            matches += 1
            calls += b["n_calls"]
            for which, m in ms.items():
                a = m.process_scan(0, init, trace=True)
                rescored[which] += m.stats()["steps_rescored"]
                n = min(a["n_calls"], b["n_calls"])
                bad = np.nonzero((a["accepted"][:n] != b["accepted"][:n]) | (a["poses"][:n] != b["poses"][:n]).any(1))[0]
                if a["n_calls"] == b["n_calls"] and len(bad) == 0:
                    np.testing.assert_allclose(a["scores"], b["scores"], rtol=1e-12, atol=0)
                    continue
                div[which] += 1
                if which == "raw":
                    i = int(bad[0]) if len(bad) else n
                    assert i < n and np.array_equal(a["poses"][i], b["poses"][i])  # same candidate, other decision
                    acc = np.nonzero(b["accepted"][:i])[0]
                    best = b["scores"][acc[-1]]
                    assert abs(b["scores"][i] - best) <= 2 * np.spacing(best), \
                        "default mode flipped a comparison that is not a tie: %r vs %r" % (b["scores"][i], best)
        assert ms["dev"].resident_stats()["gave_up"] == 0 and ms["dev"].resident_stats()["matches"] == 5
    print("fuzz over the %s OOPE: %d matches per matcher, divergences %r, re-scored steps / batches %r"
          % (oope, matches, div, rescored))
    assert matches == 5 * n_scenes and calls > matches * 60
    assert div["dev"] == 0 and div["host"] == 0, "checked default-mode matches diverged from the strict mode: %r" % div
    assert rescored["raw"] == 0 and div["raw"] <= matches // 10


def test_fuzz_default_mode_over_the_gmapping_oope(pkg, ctx, po):
    """... and over the GMapping OOPE (gmapping_occupancy_observation_pe.h:17-38).  Its per-beam value is exp() of a
    distance and the fast paths use the device's exp, so a lone matcher's default mode is checked another way (r06,
    VERDICT r5 item 2): a comparison on the walked path whose two scores lie within 2^-40 of each other is reported
    before anything has been shown to an observer, and the match is redone in the EXACT mode -- call order, beam-order
    sums, glibc's exp and the raw provider's cos / sin(theta + a) restated (csrc/libm_exact.h, exact_kernels.hip).
    The strict reference here is the ORACLE: the reference's loop with the reference's beam-order sum and libm.
      * 200 matches over 40 random scenes with failed-round limits 6 (what GMapping hard-wires, init_gmapping.h:58-60),
        10 and 14 -- the two device chains and the host-driven batches: the oracle's accept trace in every one, and
        (nearly) none of them redone: no comparison comes that close at those step sizes;
      * limit 27: steps shrink to 0.1 * 2^-27 = 7e-10 m around an optimum, where the candidates' scores differ by LESS than
        the last bits any double-precision evaluation of the sum can resolve -- r05's unchecked chains parted from the oracle
        in 17 of 20 such matches (each at a comparison within 16 ulps).  Checked: every one of them takes the oracle's path,
        scores bit-equal (they were redone in the exact mode); with the check off they part as before."""
    from synth import CELL_GMAPPING
    names = ("k1", "res", "host")
    div, div27, redone, redone27 = (dict.fromkeys(names, 0) for _ in range(4))
    matches = matches27 = calls = unchecked27 = 0
    O = po.Oracle()
    n_scenes = int(os.environ.get("SLAMHIP_FUZZ_SCENES", "40"))
    for seed in range(n_scenes + max(2, n_scenes // 10)):
        deep = seed >= n_scenes
        sc = make_scene(cell_model=CELL_GMAPPING, size=500, scale=0.05, n_beams=360 + 90 * (seed % 5), seed=500 + seed)
        upload(pkg, ctx, sc)
        ctx.scan_set_angles(sc["scan"].angle)  # (the redo then runs the reference's default raw provider, like the oracle)
        rs = np.random.RandomState(2000 + seed)
        prm = [27 if deep else (6, 6, 10, 14)[seed % 4], 0.1, 0.1]
        cfg = pkg.spe_cfg(oope=pkg.OOPE_GMAPPING)
        ms = dict(k1=pkg.Matcher(ctx, "HC", cfg, prm), res=pkg.Matcher(ctx, "HC", cfg, prm), host=pkg.Matcher(ctx, "HC", cfg, prm))
        ms["k1"].set_device_chain(1)
        ms["res"].set_device_chain(2)
        ms["host"].set_device_chain(0)
        raw = pkg.Matcher(ctx, "HC", cfg, prm)  # for the record: the same chain without the check
        raw.set_device_chain(1)
        raw.set_tie_check(0)
        for rep in range(5):
            init = sc["true_pose"] + rs.randn(3) * [0.08, 0.08, 0.04]
            b = O.process_scan(O.enumerator(po.SM_HC, prm), sc["map"], sc["scan"], po.make_cfg(oope=po.OOPE_GMAPPING), init,
                               cache=po.Oracle.new_gm_cache())
            if deep:
                matches27 += 1
                ctx.gm_cache_reset()
                a = raw.process_scan(0, init, trace=True)
                unchecked27 += int(a["n_calls"] != b["n_calls"] or not np.array_equal(a["accepted"], b["accepted"]))
            else:
                matches += 1
                calls += b["n_calls"]
            for which, m in ms.items():
                ctx.gm_cache_reset()
                a = m.process_scan(0, init, trace=True)
                was_redone = m.stats()["steps_rescored"] > 0
                (redone27 if deep else redone)[which] += int(was_redone)
                if a["n_calls"] == b["n_calls"] and np.array_equal(a["accepted"], b["accepted"]):
                    np.testing.assert_allclose(a["poses"], b["poses"], rtol=0, atol=1e-12)
                    if was_redone:  # the exact mode from the first call on, in every form: the oracle's bits
                        np.testing.assert_array_equal(a["scores"], b["scores"])
                    else:
                        np.testing.assert_allclose(a["scores"], b["scores"], rtol=1e-10, atol=1e-300)
                    continue
                (div27 if deep else div)[which] += 1
    print("fuzz over the GMapping OOPE: %d matches per matcher at limits 6 / 10 / 14: divergences from the oracle's strict loop "
          "%r, redone in the exact mode %r; %d matches at limit 27: divergences %r, redone %r (without the check %d of them part)"
          % (matches, div, redone, matches27, div27, redone27, unchecked27))
    assert matches == 5 * n_scenes and calls > matches * 40
    assert div == dict.fromkeys(names, 0), "GMapping-OOPE matches left the oracle's accept path: %r" % div
    assert div27 == dict.fromkeys(names, 0), "checked GMapping-OOPE matches at limit 27 left the oracle's accept path: %r" % div27
    assert all(v >= matches27 // 2 for v in redone27.values()) and all(v <= matches // 10 for v in redone.values())
    assert unchecked27 >= matches27 // 4  # (the check is what keeps them on the path)


@pytest.mark.parametrize("cell,weighting", [(CELL_OCC, "even"), (CELL_TBM, "viny")])
@pytest.mark.parametrize("prm", [[666666, 0.2, 0.1, 20, 100], [7, 0.2, 0.1, 4096, 4096], [11, 0.3, 0.05, 30, 1000], [5, 0.2, 0.1, 3, 2]])
@pytest.mark.parametrize("mode", CHAIN_MODES)
def test_mc_chain_equals_host_driven_matcher(pkg, ctx, cell, weighting, prm, mode):
    """The Monte-Carlo matcher on the device chain (csrc/mc_chain.hip: candidates under "all rejected" per
    super-step, first acceptance wins, the enumerator state advanced in closed form -- tape position, pending
    second Marsaglia values, failure counter, halved dispersions; mode 2: the same as ONE co-resident launch,
    csrc/mc_resident.hip) against the host-driven matcher: trace, result and the random stream left behind (the
    next match continues on the same engine), bit for bit."""
    sc = make_scene(cell_model=cell, size=600, scale=0.05, n_beams=720, seed=5, weighting=weighting)
    upload(pkg, ctx, sc)
    dev = pkg.Matcher(ctx, "MC", pkg.spe_cfg(), prm)
    host = pkg.Matcher(ctx, "MC", pkg.spe_cfg(), prm)
    dev.set_device_chain(mode)
    host.set_device_chain(0)
    for m in (dev, host):
        m.set_tie_check(0)
    init = sc["init_pose"]
    for rep in range(4):  # consecutive matches: the engine is not reseeded
        td = dev.process_scan(0, init, trace=True)
        th = host.process_scan(0, init, trace=True)
        assert_trace_equal(td, th)
        sd, sh = dev.stats(), host.stats()
        assert sd["scorer_calls"] == sh["scorer_calls"] == td["n_calls"]
        q = dev.process_scan(0, init)  # without an observer; the host twin has to draw the same numbers
        qh = host.process_scan(0, init)
        assert q["prob"] == qh["prob"] and np.array_equal(q["delta"], qh["delta"])
        init = init + np.array([0.013, -0.007, 0.004])
    assert dev.stats()["kernels_launched"] > 0 and host.stats()["kernels_launched"] == 0
    if mode == 2:
        assert dev.resident_stats() == dict(matches=8, gave_up=0) and dev.stats()["kernels_launched"] == 1
    else:
        assert dev.resident_stats() == dict(matches=0, gave_up=0)


def test_mc_resident_chain_with_1024_thread_workgroups(pkg, ctx):
    """1024-thread workgroups are resident one per CU, so the co-resident Monte-Carlo launch speculates on 252
    candidates per super-step instead of 384 (csrc/matchers.cpp): the trace, the result and the random stream left
    behind do not depend on how many candidates a super-step scores.  1080 beams: the 56 beams behind a thread's first
    one have their constants in LDS."""
    sc = make_scene(cell_model=CELL_TBM, size=600, scale=0.05, n_beams=1080, seed=8, weighting="viny")
    upload(pkg, ctx, sc)
    prm = [31, 0.2, 0.1, 600, 2000]
    dev = pkg.Matcher(ctx, "MC", pkg.spe_cfg(), prm)
    host = pkg.Matcher(ctx, "MC", pkg.spe_cfg(), prm)
    dev.set_device_chain(2, 1024)
    host.set_device_chain(0)
    init = sc["init_pose"]
    for rep in range(3):
        assert_trace_equal(dev.process_scan(0, init, trace=True), host.process_scan(0, init, trace=True))
        init = init + np.array([0.011, 0.006, -0.003])
    assert dev.resident_stats() == dict(matches=3, gave_up=0) and dev.stats()["kernels_launched"] == 1


@pytest.mark.parametrize("tie_check", [0, 1])
def test_resident_mc_chain_gives_up_when_a_workgroup_is_missing(pkg, tctx, tie_check):
    """The Monte-Carlo counterpart of the test below (csrc/mc_resident.hip): with one workgroup gone the others give up
    within the bound, nothing has been reported and the enumerator has not been touched -- the chain of kernels redoes
    the match on the same random stream; the matches after it continue on that stream as if nothing had happened."""
    ctx = tctx
    import ctypes as C
    import time
    sc = make_scene(cell_model=CELL_TBM, size=600, scale=0.05, n_beams=720, seed=5, weighting="viny")
    upload(pkg, ctx, sc)
    prm = [666666, 0.2, 0.1, 40, 300]
    dev = pkg.Matcher(ctx, "MC", pkg.spe_cfg(), prm)
    host = pkg.Matcher(ctx, "MC", pkg.spe_cfg(), prm)
    dev.set_device_chain(2)
    host.set_device_chain(0)
    for m in (dev, host):
        m.set_tie_check(tie_check)
    L = pkg.load(testing=True)
    L.slamhip_matcher_debug_resident_mute.argtypes = [C.c_void_p, C.c_int]
    L.slamhip_matcher_debug_resident_mute.restype = C.c_int
    assert_trace_equal(dev.process_scan(0, sc["init_pose"], trace=True), host.process_scan(0, sc["init_pose"], trace=True))
    assert dev.resident_stats() == dict(matches=1, gave_up=0)
    for muted, slot in enumerate((0, 5), start=1):  # candidates every super-step scores
        assert L.slamhip_matcher_debug_resident_mute(dev.h, slot + 1) == 0
        t0 = time.time()
        got = dev.process_scan(0, sc["init_pose"], trace=True)
        assert time.time() - t0 < 5.0  # bounded, not a hang
        assert_trace_equal(got, host.process_scan(0, sc["init_pose"], trace=True))
        assert dev.resident_stats() == dict(matches=1 + muted, gave_up=muted)
        assert dev.stats()["kernels_launched"] > 1  # the chain of kernels ran
    assert L.slamhip_matcher_debug_resident_mute(dev.h, 0) == 0
    assert_trace_equal(dev.process_scan(0, sc["init_pose"], trace=True), host.process_scan(0, sc["init_pose"], trace=True))
    assert dev.resident_stats() == dict(matches=4, gave_up=2) and dev.stats()["kernels_launched"] == 1


@pytest.mark.parametrize("mode", CHAIN_MODES)
def test_gmapping_oope_chain_equals_host_driven_matcher(pkg, ctx, mode):
    """The GMapping OOPE on the device chain: K3's one-pose body scores the speculation tree, the replay applies
    the reference's cross-pose cache (gmapping_occupancy_observation_pe.h:21-24,36-37,43-44) in call
    order.  Trace, result and the cache left behind must equal the host-driven matcher's bit for bit -- over
    consecutive matches (the cache is carried from one process_scan into the next)."""
    from synth import CELL_GMAPPING
    for n_beams, seed in ((360, 5), (1080, 6), (3, 7), (1, 8)):
        sc = make_scene(cell_model=CELL_GMAPPING, size=600, scale=0.05, n_beams=max(n_beams, 16), seed=seed)
        s = sc["scan"]
        if n_beams < 16:  # very short scans: runs that span the whole scan (the degenerate cache hand-over)
            keep = np.arange(n_beams) * 3
            s.range, s.angle, s.weight, s.factor = s.range[keep], s.angle[keep], np.full(n_beams, 1.0 / n_beams), s.factor[keep]
        upload(pkg, ctx, sc)
        cfg = pkg.spe_cfg(oope=pkg.OOPE_GMAPPING)
        for prm in ([6, 0.1, 0.1], [40, 0.1, 0.1]):
            dev = pkg.Matcher(ctx, "HC", cfg, prm)
            dev.set_device_chain(mode)  # 2: one co-resident launch (csrc/hc_resident_gm.hip)
            host = pkg.Matcher(ctx, "HC", cfg, prm)
            host.set_device_chain(0)
            init = sc["init_pose"]
            cache_d = cache_h = None
            for rep in range(4):
                ctx.gm_cache_reset() if rep == 0 else ctx.gm_cache_set(*cache_d)
                td = dev.process_scan(0, init, trace=True)
                cache_d = ctx.gm_cache_get()
                ctx.gm_cache_reset() if rep == 0 else ctx.gm_cache_set(*cache_h)
                th = host.process_scan(0, init, trace=True)
                cache_h = ctx.gm_cache_get()
                assert_trace_equal(td, th)
                assert cache_d == cache_h
                init = init + np.array([0.011, -0.006, 0.003])
            if n_beams >= 16:
                assert dev.stats()["launches"] <= host.stats()["launches"] or prm[0] == 6  # (super-steps vs round trips)
            if mode == 2 and n_beams >= 16:
                assert dev.resident_stats()["matches"] == 4 and dev.resident_stats()["gave_up"] == 0


@pytest.mark.parametrize("n_beams", [1089, 1100, 1137, 1138])
def test_gmapping_oope_scans_with_more_than_64_surplus_beams(pkg, ctx, po, oracle, n_beams):
    """a 1024-thread workgroup scoring 1089 .. 1137 beams gives its surplus beams to helper lanes
    (gm_score_pose_wide); r05's run resolution left the cell in front of the SECOND surplus group's first beam
    (beam 1088) unwritten, so that beam started a run -- or not -- against uninitialised LDS (Q19: a beam in its
    predecessor's cell takes the run head's value).  Every form that scores a pose with 1024 threads -- K3 wide
    (launches of at most 160 poses), the chain of kernels and the co-resident launch at 1024 threads -- against the
    ORACLE (the host-driven matcher shares the kernel): scores to 1e-11, the cross-pose cache left behind, the accept
    trace of a match.  1138 beams: the first count that takes the helper-less form (9 x surplus > 1024)."""
    from synth import CELL_GMAPPING
    # close walls: consecutive beams share end cells, so runs cross the group boundaries
    sc = make_scene(cell_model=CELL_GMAPPING, size=400, scale=0.1, n_beams=n_beams, seed=900 + n_beams)
    assert sc["scan"].n == n_beams
    upload(pkg, ctx, sc)
    cfg = pkg.spe_cfg(oope=pkg.OOPE_GMAPPING, pose_trig=1)
    ocfg = po.make_cfg(oope=po.OOPE_GMAPPING)
    rs = np.random.RandomState(n_beams)
    poses = sc["init_pose"] + rs.randn(48, 3) * [0.05, 0.05, 0.01]
    poses[1::4] = poses[0::4]  # repeated poses: the cache hands over across the whole scan
    for chunk in (48, 1, 7):  # one wide launch, lone poses, ragged calls
        ctx.gm_cache_reset()
        got = np.concatenate([ctx.score_poses(0, cfg, poses[i:i + chunk]) for i in range(0, len(poses), chunk)])
        cache = po.Oracle.new_gm_cache()
        want = oracle.score_poses(sc["map"], sc["scan"], ocfg, poses, cache)
        np.testing.assert_allclose(got, want, rtol=1e-11, atol=1e-300)
        cx, cy, pr = ctx.gm_cache_get()
        assert (cx, cy) == (cache.cx, cache.cy) and abs(pr - cache.prob) <= 1e-11 * max(pr, 1e-300)
    # how many beams sit in their predecessor's cell at the group boundary the bug was about (the test must see some)
    prm = [6, 0.1, 0.1]
    for mode, nt in ((1, 1024), (2, 1024), (2, 0), (0, 0)):
        m = pkg.Matcher(ctx, "HC", pkg.spe_cfg(oope=pkg.OOPE_GMAPPING), prm)
        m.set_device_chain(mode, nt)
        for init in (sc["init_pose"], sc["true_pose"] + np.array([0.03, -0.02, 0.01])):
            b = oracle.process_scan(oracle.enumerator(po.SM_HC, prm), sc["map"], sc["scan"], ocfg, init,
                                    cache=po.Oracle.new_gm_cache())
            ctx.gm_cache_reset()
            a = m.process_scan(0, init, trace=True)
            assert a["n_calls"] == b["n_calls"] and np.array_equal(a["accepted"], b["accepted"]), (mode, nt)
            np.testing.assert_allclose(a["scores"], b["scores"], rtol=1e-10, atol=1e-300)
            np.testing.assert_allclose(a["poses"], b["poses"], rtol=0, atol=1e-12)
        m.close()


@pytest.mark.parametrize("mode", CHAIN_MODES)
def test_trace_buffer_overflow_falls_back_to_the_host_driven_matcher(pkg, tctx, mode):
    """with an observer attached the chain writes its trace into a fixed pinned buffer; a match with more
    scorer calls than it holds (error 2) used to fail with SLAMHIP_ERR_UNSUPPORTED although the host-driven path
    has no such limit.  Now the match is redone there -- nothing has been reported at that point -- and the
    observer sees the one, complete trace.  The buffer is made artificially small through the testing hook."""
    ctx = tctx
    import ctypes as C
    sc = make_scene(cell_model=CELL_OCC, size=600, scale=0.05, n_beams=720, seed=5)
    upload(pkg, ctx, sc)
    dev, host = matchers(pkg, ctx, [32, 0.1, 0.1], mode=mode)
    L = pkg.load(testing=True)
    L.slamhip_matcher_debug_trace_cap.argtypes = [C.c_void_p, C.c_int]
    L.slamhip_matcher_debug_trace_cap.restype = C.c_int
    want = host.process_scan(0, sc["init_pose"], trace=True)
    assert want["n_calls"] > 40
    assert L.slamhip_matcher_debug_trace_cap(dev.h, 16) == 0
    got = dev.process_scan(0, sc["init_pose"], trace=True)
    assert_trace_equal(got, want)
    assert dev.stats()["kernels_launched"] > 0  # the chain did run first
    assert L.slamhip_matcher_debug_trace_cap(dev.h, 0) == 0
    again = dev.process_scan(0, sc["init_pose"], trace=True)
    assert_trace_equal(again, want)
    if mode == 2:
        # r06: ... and when the buffer ends inside the closed-form tail of the co-resident chain (limit 128: the climb's
        # ~250 calls fit, the tail's 250+ do not) -- written by the bookkeeping workgroup's threads side by side
        dev2, host2 = matchers(pkg, ctx, [128, 0.1, 0.1], mode=2)
        want2 = host2.process_scan(0, sc["init_pose"], trace=True)
        full = dev2.process_scan(0, sc["init_pose"], trace=True)
        tail = dev2.stats()["calls_closed_form"]
        assert tail > 100 and want2["n_calls"] - tail > 100
        assert L.slamhip_matcher_debug_trace_cap(dev2.h, want2["n_calls"] - tail // 2) == 0
        got2 = dev2.process_scan(0, sc["init_pose"], trace=True)
        assert_trace_equal(got2, want2)
        assert_trace_equal(full, want2)
        assert dev2.stats()["calls_closed_form"] == 0  # (the host-driven matcher made this trace)
        assert L.slamhip_matcher_debug_trace_cap(dev2.h, 0) == 0


def test_resident_chain_gives_up_when_a_workgroup_is_missing(pkg, tctx):
    """the co-resident launch only terminates when every workgroup of the tree is on the chip, so
    every wait in it is bounded.  The testing hook makes one workgroup leave at once -- what a workgroup that never
    became resident looks like: the others must give up within the bound (error 4, nothing reported), the kernel
    chain redoes the match with the same trace, and after three such matches in a row the matcher stops trying."""
    ctx = tctx
    import ctypes as C
    import time
    sc = make_scene(cell_model=CELL_OCC, size=600, scale=0.05, n_beams=720, seed=5)
    upload(pkg, ctx, sc)
    dev, host = matchers(pkg, ctx, [16, 0.1, 0.1], mode=2)
    L = pkg.load(testing=True)
    L.slamhip_matcher_debug_resident_mute.argtypes = [C.c_void_p, C.c_int]
    L.slamhip_matcher_debug_resident_mute.restype = C.c_int
    want = host.process_scan(0, sc["init_pose"], trace=True)
    assert_trace_equal(dev.process_scan(0, sc["init_pose"], trace=True), want)
    assert dev.resident_stats() == dict(matches=1, gave_up=0)
    for muted, slot in enumerate((3, 1, 200), start=1):  # a scoring workgroup each time
        assert L.slamhip_matcher_debug_resident_mute(dev.h, slot + 1) == 0
        t0 = time.time()
        got = dev.process_scan(0, sc["init_pose"], trace=True)
        assert time.time() - t0 < 5.0  # bounded: ~0.1 s of polling, not a hang
        assert_trace_equal(got, want)
        assert dev.resident_stats() == dict(matches=1 + muted, gave_up=muted)
        assert dev.stats()["kernels_launched"] > 1  # the kernel chain ran
    # three in a row: the matcher keeps to the kernel chain until told otherwise
    assert_trace_equal(dev.process_scan(0, sc["init_pose"], trace=True), want)
    assert dev.resident_stats() == dict(matches=4, gave_up=3)
    assert L.slamhip_matcher_debug_resident_mute(dev.h, 0) == 0
    dev.set_device_chain(2)
    assert_trace_equal(dev.process_scan(0, sc["init_pose"], trace=True), want)
    assert dev.resident_stats() == dict(matches=5, gave_up=3) and dev.stats()["kernels_launched"] == 1


@pytest.mark.parametrize("cell,weighting", [(CELL_OCC, "even"), (CELL_TBM, "viny")])
@pytest.mark.parametrize("oope", ["max", "mean", "overlap"])
def test_window_oopes_on_the_resident_chain_equal_the_host_driven_matcher(pkg, ctx, cell, weighting, oope):
    """VERDICT r3 item 6 (second half): the window OOPEs -- Max / Mean / OverlapWeighted
    (occupancy_observation_probability.h:29-99) -- used to take the host-driven batches with a PCIe round trip each;
    now hill climbing over them runs as the co-resident launch too (K2's per-beam value inside hc_resident.hip).
    Trace, result and scorer calls equal the host-driven matcher's bit for bit, over consecutive matches."""
    sc = make_scene(cell_model=cell, size=600, scale=0.05, n_beams=720, seed=5, weighting=weighting)
    upload(pkg, ctx, sc)
    kinds = dict(max=pkg.OOPE_MAX, mean=pkg.OOPE_MEAN, overlap=pkg.OOPE_OVERLAP)
    cfg = pkg.spe_cfg(oope=kinds[oope], area=(-0.06, 0.06, -0.04, 0.04))
    for prm in ([6, 0.1, 0.1], [40, 0.1, 0.1]):
        dev = pkg.Matcher(ctx, "HC", cfg, prm)
        host = pkg.Matcher(ctx, "HC", cfg, prm)
        host.set_device_chain(0)
        init = sc["init_pose"]
        for rep in range(3):
            td = dev.process_scan(0, init, trace=True)
            th = host.process_scan(0, init, trace=True)
            assert_trace_equal(td, th)
            assert dev.stats()["scorer_calls"] == host.stats()["scorer_calls"] == td["n_calls"]
            init = init + np.array([0.013, -0.007, 0.004])
        assert dev.resident_stats() == dict(matches=3, gave_up=0) and dev.stats()["kernels_launched"] == 1
        assert host.stats()["kernels_launched"] == 0


@pytest.mark.parametrize("cell,weighting", [(CELL_OCC, "even"), (CELL_TBM, "viny")])
@pytest.mark.parametrize("prm", [[128, 0.1, 0.1], [60, 0.1, 0.1], [1000, 0.2, 0.1], [300, 1e-9, 1e-9]])
@pytest.mark.parametrize("level,mode", [(1, 2), (2, 2), (2, 1)])  # (mode 1, the chain of kernels: identical poses only)
def test_inert_tail_ends_the_chain_in_closed_form_with_the_same_trace(pkg, ctx, po, oracle, cell, weighting, prm, level, mode):
    """r06.  Once the hill climber's steps are below half an ulp of every pose coordinate, every
    candidate of every further round IS the best pose bit for bit -- the reference goes on scoring it, 6 x (limit -
    failed) + 1 times, a tie and a rejection each time (hill_climbing_scan_matcher.h:83-101,
    pose_enumeration_scan_matcher.h:58).  The co-resident chain ends there (SLAMHIP_OPT_INERT_TAIL, default on): the
    observer still sees every one of those scorer calls -- same poses, same scores, same count -- but nothing is scored
    for them.  Against the same matcher with the option off, the host-driven matcher and the oracle's strict loop."""
    sc = make_scene(cell_model=cell, size=600, scale=0.05, n_beams=720, seed=11, weighting=weighting)
    upload(pkg, ctx, sc)
    ctx.set_option(pkg.OPT_INERT_TAIL, level)
    try:
        on = pkg.Matcher(ctx, "HC", pkg.spe_cfg(), prm)
        on.set_device_chain(mode)
        ton = [on.process_scan(0, sc["init_pose"] + k * np.array([0.011, -0.006, 0.003]), trace=True) for k in range(3)]
        son = on.stats()
        quiet = on.process_scan(0, sc["init_pose"] + 2 * np.array([0.011, -0.006, 0.003]))
        assert quiet["prob"] == ton[2]["prob"] and np.array_equal(quiet["delta"], ton[2]["delta"])
        assert on.resident_stats()["gave_up"] == 0 and (on.resident_stats()["matches"] > 0) == (mode == 2)
        ctx.set_option(pkg.OPT_INERT_TAIL, 0)
        off = pkg.Matcher(ctx, "HC", pkg.spe_cfg(), prm)
        off.set_device_chain(mode)
        toff = [off.process_scan(0, sc["init_pose"] + k * np.array([0.011, -0.006, 0.003]), trace=True) for k in range(3)]
        soff = off.stats()
    finally:
        ctx.set_option(pkg.OPT_INERT_TAIL, 2)
    host = pkg.Matcher(ctx, "HC", pkg.spe_cfg(), prm)
    host.set_device_chain(0)
    for a, b in zip(ton, toff):
        assert_trace_equal(a, b)
        assert a["n_calls"] == b["n_calls"] and a["prob"] == b["prob"] and np.array_equal(a["delta"], b["delta"])
    assert_trace_equal(ton[0], host.process_scan(0, sc["init_pose"], trace=True))
    assert son["scorer_calls"] == soff["scorer_calls"] == ton[2]["n_calls"]
    if prm[0] >= 100:  # the limit lies behind the point where the steps vanish: fewer poses scored, fewer super-steps
        assert son["poses_evaluated"] < soff["poses_evaluated"] and son["launches"] < soff["launches"]
        # ... and the trace ends in the best pose, scored as often as the reference scores it
        t, tail = ton[2], son["calls_closed_form"]
        assert tail >= 6 * (prm[0] - 100) + 1 and soff["calls_closed_form"] == 0
        assert len(set(t["scores"][-tail:])) == 1 and not np.any(t["accepted"][-tail:])
        if level == 1 or mode == 1:  # (identical poses only: the whole tail is the best pose itself)
            assert all(np.array_equal(t["poses"][-1 - q], t["poses"][-1]) for q in range(tail))
    e = oracle.enumerator(po.SM_HC, prm)
    r = oracle.process_scan(e, sc["map"], sc["scan"], po.make_cfg(), sc["init_pose"])
    assert_trace_equal(ton[0], r, exact_scores=False, rtol=1e-12)


def test_inert_tail_fuzz_against_every_call_scored(pkg, ctx):
    """160 matches over scenes, cell models, beam counts, limits and step sizes (tiny steps included: chains that are
    inert from the first round on; coarse maps and many beams: end points close to cell edges), each run at
    SLAMHIP_OPT_INERT_TAIL 2 (identical poses + the certificate, which is an argument about rounding), 1 and 0 (every
    call scored): traces, results and counts assert-equal."""
    rs = np.random.RandomState(77)
    n_closed, closed = {1: 0, 2: 0}, {1: 0, 2: 0}
    for it in range(40):
        cell = CELL_OCC if it % 2 == 0 else CELL_TBM
        scale = [0.05, 0.1, 0.025, 0.2][it % 4]
        sc = make_scene(cell_model=cell, size=[600, 400, 800, 200][it % 4], scale=scale, n_beams=[720, 360, 1080, 200][(it // 4) % 4],
                        seed=300 + it, weighting="viny" if cell == CELL_TBM else "even")
        upload(pkg, ctx, sc)
        prm = [int(rs.choice([70, 128, 200, 400])), float(rs.choice([0.3, 0.1, 0.02, 1e-12])), float(rs.choice([0.1, 0.05, 1e-3, 1e-14]))]
        poses = [sc["init_pose"] + rs.randn(3) * [0.05, 0.05, 0.02] for _ in range(4)]
        res = {}
        for level in (2, 1, 0):
            ctx.set_option(pkg.OPT_INERT_TAIL, level)
            try:
                m = pkg.Matcher(ctx, "HC", pkg.spe_cfg(), prm)
                res[level] = []
                for p in poses:
                    t = m.process_scan(0, p, trace=True)
                    res[level].append((t, m.stats()))
                assert m.resident_stats()["gave_up"] == 0
                m.close()
            finally:
                ctx.set_option(pkg.OPT_INERT_TAIL, 2)
        for lv in (2, 1):
            for (a, sa), (b, sb) in zip(res[lv], res[0]):
                assert_trace_equal(a, b)
                assert a["prob"] == b["prob"] and np.array_equal(a["delta"], b["delta"]) and a["n_calls"] == b["n_calls"]
                assert sa["scorer_calls"] == sb["scorer_calls"] and sb["calls_closed_form"] == 0
                n_closed[lv] += sa["calls_closed_form"] > 0
                closed[lv] += sa["calls_closed_form"]
    assert n_closed[1] >= 100 and n_closed[2] >= n_closed[1]  # (the shortcuts really were taken in most of them)
    assert closed[2] >= closed[1]  # (a lone chain's workgroups make no certificates: tests/test_gpu_batch.py has the batches)


def test_certificate_with_end_points_placed_next_to_cell_edges(pkg, ctx):
    """The certificate's bound, provoked: a map of random occupancies (every cell differs from its neighbours, so a
    beam that changes cells changes the score), scans of 1 ... 12 beams whose end points are PLACED at a distance
    1e-13 ... 1e-3 m from a cell edge under the initial pose -- on either side, in x or in y, at ranges up to 25 m --
    and hill climbers that start with steps of 1e-2 ... 1e-7, i.e. on either side of that distance: which candidates
    cross the edge, and are accepted, depends on the last digits of the geometry.  A certificate made for the wrong
    pose or the wrong steps, or one that held too early, would end a chain where the reference goes on to accept: 300
    matches at SLAMHIP_OPT_INERT_TAIL 2 and 0, traces assert-equal; the shortcut is taken in most of them."""
    from synth import MapData
    rs = np.random.RandomState(2024)
    scale = 0.1
    m = MapData(CELL_OCC, rs.rand(300, 300), (150, 150), scale, [0.5])
    ctx.upload_map(0, m)
    n_closed = n_acc_late = 0
    for it in range(300):
        nb = int(rs.choice([1, 2, 3, 6, 12]))
        pose = np.array([rs.uniform(-3, 3), rs.uniform(-3, 3), rs.uniform(-3.1, 3.1)])
        ang = rs.uniform(-2.3, 2.3, nb)
        rng = rs.uniform(0.5, 12.0, nb)
        if it % 3 == 0:
            rng[0] = rs.uniform(10.0, 25.0)  # (a long beam: the rotation candidates' lever)
        # beam 0 (and beam 1, if there is one): the end point a distance d from a cell edge, on either side
        for b in range(min(nb, 2)):
            d = 10.0 ** rs.uniform(-13, -3) * rs.choice([-1.0, 1.0])
            c, s = np.cos(pose[2] + ang[b]), np.sin(pose[2] + ang[b])
            if rs.rand() < 0.5 and abs(c) > 0.2:
                edge = np.round((pose[0] + rng[b] * c) / scale) * scale
                rng[b] = (edge + d - pose[0]) / c
            elif abs(s) > 0.2:
                edge = np.round((pose[1] + rng[b] * s) / scale) * scale
                rng[b] = (edge + d - pose[1]) / s
            rng[b] = abs(rng[b])
        cos_a, sin_a = pkg.beam_trig(ang)
        ctx.scan_upload(rng, cos_a, sin_a, np.full(nb, 1.0 / nb), np.ones(nb))
        prm = [int(rs.choice([60, 128])), 10.0 ** rs.uniform(-7, -2), 10.0 ** rs.uniform(-7, -2)]
        res = {}
        for level in (2, 0):
            ctx.set_option(pkg.OPT_INERT_TAIL, level)
            try:
                mm = pkg.Matcher(ctx, "HC", pkg.spe_cfg(), prm)
                res[level] = (mm.process_scan(0, pose, trace=True), mm.stats())
                assert mm.resident_stats()["gave_up"] == 0
                mm.close()
            finally:
                ctx.set_option(pkg.OPT_INERT_TAIL, 2)
        (a, sa), (b_, sb) = res[2], res[0]
        assert_trace_equal(a, b_)
        assert a["prob"] == b_["prob"] and np.array_equal(a["delta"], b_["delta"]) and a["n_calls"] == b_["n_calls"]
        n_closed += sa["calls_closed_form"] > 0
        n_acc_late += int(np.count_nonzero(a["accepted"]) > 0)  # (a candidate did cross an edge and was accepted)
    assert n_closed >= 200 and n_acc_late >= 60
