import pickle, heapq, sys
import numpy as np
from collections import defaultdict
seqs = pickle.load(open("/tmp/hc_seqs.pkl", "rb"))
train = [s for i, s in enumerate(seqs) if (i // 2) % 2 == 0]
test = [s for i, s in enumerate(seqs) if (i // 2) % 2 == 1]

def build(prob_fn, ctx, max_inst, min_reach=0.0):
    """prob_fn(history tuple) -> 7 probs. returns tree as dict node-> children dict"""
    heap = [(-1.0, 0, ctx, None, None)]
    nodes = []  # (children dict)
    cnt = 1
    while heap and len(nodes) < max_inst:
        pr, _, hist, parent, out = heapq.heappop(heap)
        pr = -pr
        if nodes and pr < min_reach: break
        idx = len(nodes); nodes.append({})
        if parent is not None: nodes[parent][out] = idx
        p = prob_fn(hist)
        for o in range(7):
            if p[o] <= 0: continue
            heapq.heappush(heap, (-pr * p[o], cnt, (hist + (o,))[-3:], idx, o)); cnt += 1
    return nodes

def steps_for(seq, shape_for_ctx):
    t, steps = 0, 0
    hist = ()
    n = len(seq)
    while t < n:
        nodes = shape_for_ctx(hist)
        node = 0
        while True:
            o = seq[t]; t += 1; hist = (hist + (o,))[-3:]
            if t >= n: break
            nxt = nodes[node].get(o)
            if nxt is None: break
            node = nxt
        steps += 1
    return steps

def evaluate(name, shape_for_ctx):
    tot = {6: [0, 0], 128: [0, 0]}
    for lim, outs, tail in test:
        s = steps_for(outs, shape_for_ctx)
        tot[lim][0] += s; tot[lim][1] += 1
    print("%-40s HC128: %.2f steps/match   HC6: %.2f" % (name, tot[128][0] / tot[128][1], tot[6][0] / tot[6][1]))

def iid(p):
    q = 1 - p
    v = [q ** 6] + [p * q ** (6 - j) for j in range(1, 7)]
    s = sum(v); return [x / s for x in v]

for R in (64, 128):
    for p in (0.03, 0.06, 0.12, 0.25):
        sh = build(lambda h: iid(p), (), R)
        evaluate("iid p=%.2f R=%d" % (p, R), lambda h: sh)
# markov order 1 and 2 from training data
for order in (1, 2, 3):
    cnt = defaultdict(lambda: np.ones(7) * 0.05)
    for lim, outs, tail in train:
        h = ()
        for o in outs:
            cnt[h[-order:] if order else ()][o] += 1
            h = (h + (o,))[-3:]
    def pf(h, cnt=cnt, order=order):
        c = cnt[h[-order:]]
        return list(c / c.sum())
    for R in (64, 128):
        cache = {}
        def sfc(h, R=R, pf=pf, cache=cache, order=order):
            k = h[-order:]
            if k not in cache: cache[k] = build(pf, k, R)
            return cache[k]
        evaluate("markov-%d R=%d" % (order, R), sfc)

# ---- DAG shapes: states reached by different orders of moves on different axes are ONE node
def state_key(path):
    """path: tuple of outcomes. per-axis sequences of (scale, sign) + number of halvings"""
    h = 0
    ax = ([], [], [])
    for o in path:
        if o == 0:
            h += 1
        else:
            c = o - 1
            ax[c % 3].append((h, c % 2))
    return (tuple(ax[0]), tuple(ax[1]), tuple(ax[2]), h)

def build_dag(prob7, max_nodes):
    # best-first over states; a state's priority = summed probability of the paths found so far
    heap = [(-1.0, 0, ())]
    nodes = {}   # key -> dict(out -> key)
    order = []
    pending = {}  # key -> prob accumulated while waiting
    cnt = 1
    rep = {}  # key -> representative path
    while heap and len(order) < max_nodes:
        pr, _, path = heapq.heappop(heap)
        k = state_key(path)
        if k in nodes:
            continue
        nodes[k] = {}
        rep[k] = path
        order.append(k)
        for o in range(7):
            p = prob7[o]
            heapq.heappush(heap, (pr * p, cnt, path + (o,))); cnt += 1
    # link children that exist
    for k in order:
        for o in range(7):
            ck = state_key(rep[k] + (o,))
            if ck in nodes:
                nodes[k][o] = ck
    return nodes, order[0]

def steps_dag(seq, nodes, root):
    t, steps, n = 0, 0, len(seq)
    while t < n:
        path = ()
        node = root
        while True:
            o = seq[t]; t += 1
            if t >= n: break
            nxt = nodes[node].get(o)
            if nxt is None: break
            node = nxt
        steps += 1
    return steps

for R in (64, 128):
    for p in (0.03, 0.06, 0.12):
        nodes, root = build_dag(iid(p), R)
        tot = {6: [0, 0], 128: [0, 0]}
        for lim, outs, tail in test:
            s = steps_dag(outs, nodes, root)
            tot[lim][0] += s; tot[lim][1] += 1
        print("DAG iid p=%.2f R=%d   HC128: %.2f steps/match   HC6: %.2f" % (p, R, tot[128][0] / tot[128][1], tot[6][0] / tot[6][1]))
