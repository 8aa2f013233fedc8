"""CPU suite, world_size 2 and 3 over gloo: the sharded particle-filter bookkeeping.

Each rank owns a contiguous block of the particles (uneven when the count does not divide: the first
ranks hold one more, as bench.py shards 100 particles over 8 GPUs); the only exchange is the
all-gather of the raw weights (and, when a resampling happens, of the particle records).  Every rank
must take the same decision, compute the same indices, and end up with exactly the particles an
unsharded filter holds.  The GPU matching itself is covered by the -m gpu tests; here the filters are
created without a context (host-only) and the scan probabilities are injected."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def split(n, world):
    counts = [n // world + (1 if r < n % world else 0) for r in range(world)]
    return counts, [sum(counts[:r]) for r in range(world)]


def scenario(step, n):
    rs = np.random.RandomState(100 + step)
    probs = rs.rand(n) ** 3 + 1e-3  # wide spread -> N_eff collapses -> resampling
    poses = rs.randn(n, 3)
    return probs, poses


def run_steps(pkg, n, first, count, gather):
    gp = [0.0, 0.1, 0.0, 0.03, 0.0, 0.0, 0.0, 0.0]
    seeds = np.arange(1000, 1000 + n, dtype=np.uint32)[first:first + count]
    pf = pkg.GmappingFilter(None, pkg.gmapping_params(gp8=gp), n, seeds, first=first, count=count)
    log = []
    for step in range(4):
        probs, poses = scenario(step, n)
        _, w, _ = pf.state()
        # what predict_match would leave behind: new poses, weight *= scan probability
        pf.set(poses=poses[first:first + count], weights=w * probs[first:first + count])
        _, raw, _ = pf.state()
        all_raw = gather(raw)
        # the travelled-distance gate stays shut without odometry, so the resampling is driven through
        # import directly, with the same N_eff rule on every rank
        req, idx = pf.plan_resample(all_raw, 7 + step)
        wn = all_raw / all_raw.sum()
        need = 2.0 / np.sum(wn * wn) < n
        if need:  # UniformResamling::resampling_is_required, applied identically on all ranks
            idx = pkg.pf_resample(pkg.pf_normalize(all_raw), 7 + step)
            blobs = gather(pf.export())
            pf.import_(blobs, idx)
        poses_l, w_l, ms_l = pf.state()
        log.append((need, idx.copy() if need else None, poses_l, w_l, ms_l))
    return log


def worker(rank, world, port, q, n):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    pkg = ge.load_package()
    counts, firsts = split(n, world)
    count, first = counts[rank], firsts[rank]

    def gather(a):  # per-particle rows, padded to the largest shard
        a = np.ascontiguousarray(a)
        per = a.size // count
        padded = np.zeros(max(counts) * per, dtype=a.dtype)
        padded[:a.size] = a.ravel()
        t = torch.from_numpy(padded)
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        return np.concatenate([outs[r].numpy()[:counts[r] * per] for r in range(world)])

    log = run_steps(pkg, n, first, count, gather)
    q.put((rank, [(nd, None if i is None else i.tolist(), p.tolist(), w.tolist(), m.tolist())
                  for nd, i, p, w, m in log]))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 24), (3, 25)])
def test_shards_equal_unsharded_filter(world, n):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    pkg = ge.load_package()
    if not os.path.exists(pkg.LIB_PATH):
        pkg.build()
    ref_log = run_steps(pkg, n, 0, n, lambda a: np.asarray(a))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 7 * world) % 2000
    procs = [ctx.Process(target=worker, args=(r, world, port, q, n)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert any(s[0] for s in ref_log), "scenario never resampled"
    counts, firsts = split(n, world)
    for step, (need, idx, poses, w, ms) in enumerate(ref_log):
        for rank in range(world):
            n_, i_, p_, w_, m_ = got[rank][step]
            lo, hi = firsts[rank], firsts[rank] + counts[rank]
            assert n_ == need
            if need:
                assert i_ == idx.tolist()  # identical indices on every rank
            np.testing.assert_array_equal(np.array(p_), poses[lo:hi])
            np.testing.assert_array_equal(np.array(w_), w[lo:hi])
            np.testing.assert_array_equal(np.array(m_), ms[lo:hi])
    # exactly one master overall after every step
    for step in range(len(ref_log)):
        assert sum(sum(got[r][step][4]) for r in range(world)) == 1
