"""Simulate slack-aware level assignment of the batch's tree nodes (diagnostic for the scheduler)."""
import sys, os, heapq
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vqa_amd import synth, vocab as V
from vqa_amd.schedule import Plan

def cost(n, full=46.0, half=30.0):
    r, rem = divmod(n, 512)
    return r * full + (0 if rem == 0 else half if rem <= 256 else full)

def balance(nodes):
    NV = len(nodes)
    L = int(nodes[:, 5].max()) + 1
    parent = np.full(NV, -1, dtype=np.int64)
    nchild = np.zeros(NV, dtype=np.int64)
    for v in range(NV):
        for c in (nodes[v, 3], nodes[v, 4]):
            if c >= 0:
                parent[c] = v
                nchild[v] += 1
    alap = np.full(NV, L - 1, dtype=np.int64)
    for v in range(NV - 1, -1, -1):
        if parent[v] >= 0:
            alap[v] = alap[parent[v]] - 1
    pending = nchild.copy()
    ready = [(int(alap[v]), v) for v in range(NV) if nchild[v] == 0]
    heapq.heapify(ready)
    done, sizes = 0, []
    for t in range(L):
        remaining, steps_left = NV - done, L - t
        share = -(-remaining // steps_left)
        target = 256 if share <= 256 else -(-share // 512) * 512
        take = []
        while ready and (ready[0][0] <= t or len(take) < target):
            take.append(heapq.heappop(ready)[1])
        for v in take:
            p = parent[v]
            if p >= 0:
                pending[p] -= 1
                if pending[p] == 0:
                    heapq.heappush(ready, (int(alap[p]), int(p)))
        # NOTE: a parent pushed here is only popped in a later step because the pops above are over
        done += len(take)
        sizes.append(len(take))
    assert done == NV, (done, NV)
    return sizes

vocab = V.make_vocab()
for seed in (3000, 3100, 3200):
    prog = synth.synth_programs(seed, 1024, vocab)
    nodes = Plan(prog, vocab).nodes()
    asap = np.bincount(nodes[:, 5])
    sizes = balance(nodes)
    print('ASAP %d levels cost %.0f us | balanced cost %.0f us  %s' % (len(asap), sum(cost(n) for n in asap), sum(cost(n) for n in sizes), sizes))
