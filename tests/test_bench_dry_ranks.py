"""CPU suite: `python bench.py --dry-ranks N` -- the benchmark's multi-rank control flow without a GPU (the same
self-launch as --gpus N: torch.distributed.run child, rendezvous on 127.0.0.1; gloo; host-only filter shards; block
split, all-gathers, identical resampling on every rank, the map-migration plan with dummy maps point to point,
barrier + max-over-ranks timing), every rank checked against an unsharded filter.  N = 8 is the node the driver's
scaling run uses: 100 particles in blocks of 13 x 4 + 12 x 4."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("ranks,particles", [(2, 24), (8, 100)])
def test_dry_ranks(ranks, particles):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-ranks", str(ranks), "--particles",
                        str(particles), "--pf-steps", "5"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [x for x in r.stdout.splitlines() if x.startswith("{")][-1]
    d = json.loads(line)
    assert d["dry_run"] and d["ok"] and d["ranks"] == ranks and sum(d["shards"]) == particles
    assert d["ranks_that_disagree_with_the_unsharded_filter"] == 0
    assert d["resamplings"] >= 1 and d["dummy_map_bytes_moved"] > 0
    # (r05: what the sharded per-particle-maps leg -- the default at --gpus N > 1 -- reports from the library's counters)
    assert d["maps_migrated"] >= 1 and d["p2p_exchanges_max_over_ranks"] >= 1
    assert d["dummy_map_bytes_moved"] == 64 * d["maps_migrated"]
    if ranks == 8:
        assert d["shards"] == [13] * 4 + [12] * 4
    # r06: the WEAK-scaling form beside the strong one -- `--particles` particles per rank -- walked
    # through the same control flow, every rank checked against the unsharded filter of ranks x particles
    w = d["weak"]
    assert w["ranks"] == ranks and w["particles"] == ranks * particles and w["shards"] == [particles] * ranks
    assert w["ok"] and w["ranks_that_disagree_with_the_unsharded_filter"] == 0 and w["resamplings"] >= 1
    for leg in (d, w):
        assert leg["ms_per_step"] > 0 and leg["particles_per_s"] > 0
