"""GPU suite: two contexts on ONE device in the default mode (SURVEY 8e "replicas": one world per
robot, src/core/states/world.h:9-11).  A co-resident launch only runs when all its workgroups are on the chip, and a
lone hill-climbing chain asks for the whole device -- two of them launched at once could each get half and wait for
the other until their spin bounds ran out.  The library's per-device ledger of resident slots (csrc/matchers.cpp)
gives the form to one caller at a time; the other runs the same match as the chain of kernels AT ONCE.  Asserted: the
results of two threads matching concurrently equal their lone runs bit for bit, nobody gives up, and the pair is not
slower than three times the two runs one after the other."""
import threading
import time

import numpy as np
import pytest
from synth import CELL_OCC, CELL_TBM, make_scene

import __graft_entry__ as ge

pytestmark = pytest.mark.gpu
N_MATCHES = 200


def robot(pkg, seed, cell, weighting, kind):
    ctx = pkg.Context(0)
    sc = make_scene(cell_model=cell, size=600, scale=0.05, n_beams=720, seed=seed, weighting=weighting)
    ctx.upload_map(0, sc["map"])
    c, s = pkg.beam_trig(sc["scan"].angle)
    ctx.scan_upload(sc["scan"].range, c, s, sc["scan"].weight, sc["scan"].factor)
    prm = [32, 0.1, 0.1] if kind == "HC" else [7 + seed, 0.2, 0.1, 60, 600]
    m = pkg.Matcher(ctx, kind, pkg.spe_cfg(), prm)  # the default mode: co-resident launch, checked
    rs = np.random.RandomState(seed)
    inits = [sc["true_pose"] + rs.randn(3) * [0.06, 0.06, 0.03] for _ in range(N_MATCHES)]
    return dict(ctx=ctx, m=m, inits=inits)


def run(r, out):
    t0 = time.perf_counter()
    res = []
    for p in r["inits"]:
        q = r["m"].process_scan(0, p)
        res.append((q["prob"], tuple(q["delta"]), r["m"].stats()["scorer_calls"]))
    out.append((res, time.perf_counter() - t0))


@pytest.mark.parametrize("kinds", [("HC", "HC"), ("HC", "MC")])
def test_two_contexts_share_one_device(kinds):
    pkg = ge.load_package()
    robots = [robot(pkg, 21, CELL_OCC, "even", kinds[0]), robot(pkg, 22, CELL_TBM if kinds[1] == "MC" else CELL_OCC,
                                                                "viny" if kinds[1] == "MC" else "even", kinds[1])]
    # lone runs, one after the other (a Monte-Carlo matcher's engine runs on: a fresh matcher per pass below)
    lone = []
    for r in robots:
        out = []
        run(r, out)
        lone.append(out[0])
    assert all(r["m"].resident_stats()["matches"] == N_MATCHES and r["m"].resident_stats()["gave_up"] == 0 for r in robots)
    # the same matches, both robots at once
    fresh = [robot(pkg, 21, CELL_OCC, "even", kinds[0]), robot(pkg, 22, CELL_TBM if kinds[1] == "MC" else CELL_OCC,
                                                               "viny" if kinds[1] == "MC" else "even", kinds[1])]
    outs = [[], []]
    th = [threading.Thread(target=run, args=(fresh[i], outs[i])) for i in range(2)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    wall = time.perf_counter() - t0
    for i in range(2):
        assert outs[i][0][0] == lone[i][0], "robot %d: concurrent results differ from its lone run" % i
    stats = [r["m"].resident_stats() for r in fresh]
    seq = lone[0][1] + lone[1][1]
    print("two contexts (%s, %s): lone %.1f + %.1f ms, together %.1f ms wall; co-resident launches / give-ups %r"
          % (kinds[0], kinds[1], 1e3 * lone[0][1], 1e3 * lone[1][1], 1e3 * wall, stats))
    assert all(s["gave_up"] == 0 for s in stats), stats
    assert stats[0]["matches"] + stats[1]["matches"] >= N_MATCHES // 2  # the form is still used: by whoever gets the slots
    assert wall <= 3.0 * seq
    for r in robots + fresh:
        r["m"].close()
        r["ctx"].close()
