"""GPU suite: the probability plane of a TBM map (csrc/slamhip_internal.h DeviceMap::d_prob).

A TBM cell's per-beam probability under the discrepancy OIE is a pure function of the cell -- the scorer's observation is
always (u, e, o, c) = (0, 0, 1, 0) (src/core/maps/tbm_grid_cells.h:21-35, transferable_belief_model.h:102-143; SURVEY
Q18) -- so the 1-cell scorers gather it from an 8-byte plane instead of evaluating the belief arithmetic per (pose,
beam).  The plane is made of the SAME operations (tbm_discrepancy_probability, one definition for the scorers, the
plane's writers and the host), so:
  * every scorer form returns the same bits with the plane on and off (SLAMHIP_OPT_TBM_PLANE) -- K1 in every sum order and
    trig mode, the hill-climbing chain in its two device forms and host-driven, Monte Carlo, brute force;
  * after every kind of writer (full and partial uploads, the dirty log, the three K6 pipelines with the TBM rule, a
    window that grows) the stored plane equals the derived one, bit for bit (the testing library's
    slamhip_map_debug_prob_plane) and the scores equal a fresh upload's."""
import ctypes as C

import numpy as np
import pytest
from helpers import assert_trace_equal
from synth import CELL_TBM, make_scene

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


def plane(ctx, map_id):
    valid, bad = C.c_int(-1), C.c_longlong(-1)
    assert ctx.L.slamhip_map_debug_prob_plane(ctx.h, map_id, C.byref(valid), C.byref(bad)) == 0
    return valid.value, bad.value


def upload(pkg, ctx, sc, map_id=0):
    ctx.upload_map(map_id, sc["map"])
    c, s = pkg.beam_trig(sc["scan"].angle)
    ctx.scan_upload(sc["scan"].range, c, s, sc["scan"].weight, sc["scan"].factor)
    ctx.scan_set_angles(sc["scan"].angle)


def both(pkg, ctx, fn):
    """fn() with the plane on and off"""
    out = []
    for on in (1, 0):
        ctx.set_option(pkg.OPT_TBM_PLANE, on)
        out.append(fn())
    ctx.set_option(pkg.OPT_TBM_PLANE, 1)
    return out


def test_every_scorer_form_returns_the_same_bits_through_the_plane(pkg, tctx):
    ctx = tctx
    sc = make_scene(cell_model=CELL_TBM, size=600, scale=0.05, n_beams=720, seed=5, weighting="viny")
    upload(pkg, ctx, sc)
    rs = np.random.RandomState(4)
    poses = sc["init_pose"] + rs.randn(300, 3) * [0.2, 0.2, 0.1]
    poses[7] = [400.0, 400.0, 0.3]  # every end point outside the window: the prototype cell's probability
    for kw in (dict(), dict(sum_order=1), dict(pose_trig=1), dict(sum_order=1, pose_trig=1), dict(pose_trig=2),
               dict(sum_order=1, pose_trig=2)):
        n = 24 if kw.get("pose_trig") == 2 else len(poses)
        on, off = both(pkg, ctx, lambda: ctx.score_poses(0, pkg.spe_cfg(**kw), poses[:n]))
        np.testing.assert_array_equal(on, off, err_msg=repr(kw))
    assert plane(ctx, 0) == (1, 0)
    for kind, prm in (("HC", [64, 0.1, 0.1]), ("MC", [666666, 0.2, 0.1, 200, 1000]), ("BF", [-0.2, 0.2, 0.05, -0.2, 0.2, 0.05, -0.05, 0.05, 0.01])):
        for mode in ((2, 1, 0) if kind != "BF" else (None,)):
            def run():
                m = pkg.Matcher(ctx, kind, pkg.spe_cfg(), prm)
                if mode is not None:
                    m.set_device_chain(mode)
                t = m.process_scan(0, sc["init_pose"], trace=True)
                m.close()
                return t
            on, off = both(pkg, ctx, run)
            assert_trace_equal(on, off)
            assert on["n_calls"] > 20


def test_the_plane_follows_every_writer(pkg, tctx):
    ctx = tctx
    sc = make_scene(cell_model=CELL_TBM, size=500, scale=0.05, n_beams=720, seed=8, weighting="viny")
    m, scan = sc["map"], sc["scan"]
    upload(pkg, ctx, sc, map_id=3)
    ctx.map_set_auto_grow(3, True)
    cfg = pkg.spe_cfg()
    rs = np.random.RandomState(9)
    poses = sc["init_pose"] + rs.randn(64, 3) * [0.2, 0.2, 0.1]
    assert plane(ctx, 3)[0] == 0  # nobody has asked yet
    ctx.score_poses(3, cfg, poses)
    assert plane(ctx, 3) == (1, 0)

    def check(what):
        assert plane(ctx, 3) == (1, 0), what
        on, off = both(pkg, ctx, lambda: ctx.score_poses(3, cfg, poses))
        np.testing.assert_array_equal(on, off, err_msg=what)

    # a partial upload: other belief masses in a block of cells
    blk = np.tile(np.array([0.2, 0.1, 0.6, 0.1]), (40, 50, 1)) + rs.rand(40, 50, 4) * 0.05
    ctx.map_upload_window(3, 100, 120, blk)
    check("partial upload")
    # the dirty log
    xy = rs.randint(0, 500, (300, 2)).astype(np.int32)
    vals = rs.dirichlet([1, 1, 1, 1], 300)
    ctx.map_apply_dirty(3, xy, vals)
    check("dirty log")
    # K6 with the TBM rule, every pipeline, from several poses
    c, s = pkg.beam_trig(scan.angle)
    for path in (0, 1, 2):
        ctx.set_option(pkg.OPT_K6_PATH, path)
        for k in range(2):
            pose = sc["true_pose"] + rs.randn(3) * [0.2, 0.2, 0.05]
            nu = ctx.map_append_scan(3, pkg.RULE_TBM, pose, scan.range, c, s, None, quality=0.9, base=(0.95, 0.04, 0.01, 0.003), blur=0.1)
            assert nu > 1000
            check("K6 path %d scan %d" % (path, k))
    ctx.set_option(pkg.OPT_K6_PATH, 0)
    ctx.map_append_scan_raw(3, pkg.RULE_TBM, sc["true_pose"], scan.range, scan.angle, None, quality=0.9,
                            base=(0.95, 0.04, 0.01, 0.003))
    check("K6 with the raw provider")
    # a window that grows drops the plane; the next scorer call derives it again
    grown0 = ctx.map_info(3)["times_grown"]
    far = np.array([sc["true_pose"][0] + 9.0, sc["true_pose"][1], 0.0])
    ctx.map_append_scan(3, pkg.RULE_TBM, far, np.full(16, 6.0), np.cos(np.linspace(-1, 1, 16)), np.sin(np.linspace(-1, 1, 16)), None,
                        base=(0.95, 0.04, 0.01, 0.003))
    assert ctx.map_info(3)["times_grown"] > grown0 and plane(ctx, 3)[0] == 0
    check_scores = ctx.score_poses(3, cfg, poses)
    assert plane(ctx, 3) == (1, 0)
    ctx.set_option(pkg.OPT_TBM_PLANE, 0)
    np.testing.assert_array_equal(check_scores, ctx.score_poses(3, cfg, poses))
    ctx.set_option(pkg.OPT_TBM_PLANE, 1)
    ctx.map_release(3)
