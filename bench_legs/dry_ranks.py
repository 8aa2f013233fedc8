"""bench_legs.dry_ranks -- `bench.py --dry-ranks N`: the benchmark's multi-rank control flow without a GPU."""
import json
import os
import sys
import time

import numpy as np


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
    # the two forms `bench.py --gpus N` runs the filter in: STRONG -- --particles particles over the
    # ranks -- and WEAK -- --particles particles PER rank
    strong = _walk(args, pkg, torch, dist, rank, world, args.particles)
    weak = _walk(args, pkg, torch, dist, rank, world, args.particles * world)
    if rank == 0:
        out = dict(strong)
        out["weak"] = weak
        out["ok"] = bool(strong["ok"] and weak["ok"])
        print(json.dumps(out))
    dist.destroy_process_group()
    sys.exit(0 if strong["ranks_that_disagree_with_the_unsharded_filter"] == 0 and
             weak["ranks_that_disagree_with_the_unsharded_filter"] == 0 else 1)


def _walk(args, pkg, torch, dist, rank, world, n):
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
        log, moved, received, exchanges = [], 0, 0, 0
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
                        exchanges += 1
                        for wk in dist.batch_isend_irecv(ops):
                            wk.wait()
                    received += len(recv)
                    new = {}
                    for j in range(first_, first_ + count_):
                        src = int(idx[j])
                        new[j] = maps[src] if owner(src) == rank else recv[src].numpy().tobytes()
                    maps = new
            p_, w_, m_ = pf.state()
            log.append((need, idx, p_, w_, m_, dict(maps)))
        pf.close()
        return log, (moved, received, exchanges)

    dummy = lambda j: ((np.arange(64, dtype=np.int64) * 3 + 7 * j + j // 251) % 256).astype(np.uint8).tobytes()  # noqa: E731
    dist.barrier()
    t0 = time.perf_counter()
    log, (moved, received, exchanges) = run(first, count, gather, {j: dummy(j) for j in range(first, first + count)})
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
    tt = torch.tensor([dt, float(moved), 0.0 if ok else 1.0, float(received), float(exchanges)], dtype=torch.float64)
    mx = tt.clone()
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    sm = tt.clone()
    dist.all_reduce(sm, op=dist.ReduceOp.SUM)
    return {"dry_run": True, "ranks": world, "particles": n, "shards": counts, "steps": args.pf_steps,
            "resamplings": resamplings, "dummy_map_bytes_moved": sm[1].item(),
            "maps_migrated": int(sm[3].item()), "p2p_exchanges_max_over_ranks": int(mx[4].item()),
            "ranks_that_disagree_with_the_unsharded_filter": int(sm[2].item()),
            "ok": bool(sm[2].item() == 0 and resamplings > 0), "seconds": mx[0].item(),
            "ms_per_step": 1e3 * mx[0].item() / max(args.pf_steps, 1), "particles_per_s": n * args.pf_steps / mx[0].item(),
            "note": "host-only filter shards over gloo, launched like --gpus N; no GPU touched"}


