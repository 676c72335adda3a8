"""What a nested-dissection multifrontal factorisation of a mid-size pose graph would look like (host-only probe).

Rows in Hilbert order of the initial poses (the row plan's order), index bisection, separator = greedy vertex cover of the
cut, recursion to leaves of <= LEAF rows; symbolic factorisation bottom-up.  Prints per tree level: fronts, own / boundary
sizes (in 3x3 blocks), flops of the largest front and of the level -- the figures that decide whether one workgroup per
front is enough ("mid-size regime").

    python scripts/nd_probe.py C2 [leaf]
"""
import sys

import numpy as np

sys.path.insert(0, ".")
from sparse_gslam_amd import synth  # noqa: E402


def hilbert_key(x, y, bits=16):
    x = np.asarray(x, dtype=np.int64).copy()
    y = np.asarray(y, dtype=np.int64).copy()
    d = np.zeros_like(x)
    s = 1 << (bits - 1)
    while s > 0:
        rx = (x & s) > 0
        ry = (y & s) > 0
        d += s * s * ((3 * rx.astype(np.int64)) ^ ry.astype(np.int64))
        sw = ~ry
        fl = sw & rx
        x = np.where(fl, s - 1 - x, x)
        y = np.where(fl, s - 1 - y, y)
        x2 = np.where(sw, y, x)
        y2 = np.where(sw, x, y)
        x, y = x2, y2
        s >>= 1
    return d


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "C2"
    leaf = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    g = synth.config(name)
    V = g.V
    free = ~g.fixed.astype(bool)
    p = g.poses
    lo, hi = p[:, :2].min(0), p[:, :2].max(0)
    q = ((p[:, :2] - lo) / np.maximum(hi - lo, 1e-9) * 65535).astype(np.int64)
    key = hilbert_key(q[:, 0], q[:, 1])
    order = np.argsort(key, kind="stable")
    order = order[free[order]]
    n = len(order)
    pos = -np.ones(V, dtype=np.int64)
    pos[order] = np.arange(n)
    adj = [set() for _ in range(n)]
    for a, b in zip(pos[g.ei], pos[g.ej]):
        if a >= 0 and b >= 0 and a != b:
            adj[a].add(int(b))
            adj[b].add(int(a))

    # tree nodes: (own vertex list, children)
    nodes = []

    def dissect(verts, depth):
        if len(verts) <= leaf:
            nodes.append(dict(own=list(verts), kids=[], depth=depth))
            return len(nodes) - 1
        verts = sorted(verts)
        half = len(verts) // 2
        A, B = set(verts[:half]), set(verts[half:])
        # greedy vertex cover of the cut edges
        cut = {}
        for a in A:
            for b in adj[a]:
                if b in B:
                    cut.setdefault(a, set()).add(b)
                    cut.setdefault(b, set()).add(a)
        sep = set()
        while cut:
            v = max(cut, key=lambda u: len(cut[u]))
            sep.add(v)
            for u in cut.pop(v):
                cut[u].discard(v)
                if not cut[u]:
                    del cut[u]
        A -= sep
        B -= sep
        kids = []
        if A:
            kids.append(dissect(A, depth + 1))
        if B:
            kids.append(dissect(B, depth + 1))
        nodes.append(dict(own=sorted(sep), kids=kids, depth=depth))
        return len(nodes) - 1

    sys.setrecursionlimit(10000)
    root = dissect(set(range(n)), 0)
    # elimination position: post-order = order of creation
    elim = np.zeros(n, dtype=np.int64)
    node_of = np.zeros(n, dtype=np.int64)
    c = 0
    for i, nd in enumerate(nodes):
        for v in nd["own"]:
            elim[v] = c
            node_of[v] = i
            c += 1
    assert c == n
    # symbolic: boundary(node) = (adj(own) u children's boundaries) with elim > own's last
    for i, nd in enumerate(nodes):
        own = set(nd["own"])
        bd = set()
        for v in own:
            for u in adj[v]:
                if u not in own and node_of[u] > i:
                    bd.add(u)
        for k in nd["kids"]:
            bd |= nodes[k]["bnd"] - own
        nd["bnd"] = bd
    # height from the leaves (the level schedule)
    for nd in nodes:
        nd["h"] = 1 + max((nodes[k]["h"] for k in nd["kids"]), default=-1)
    H = nodes[root]["h"]
    tot_flops = 0.0
    tot_mem = 0
    print(f"{name}: n={n} leaf={leaf} nodes={len(nodes)} height={H}")
    print(" h  fronts  own(max/mean)  bnd(max/mean)  front dims max   MFMA-flops max front   level Mflop   pivots(max)")
    crit = 0.0
    for h in range(H + 1):
        L = [nd for nd in nodes if nd["h"] == h]
        own = np.array([len(nd["own"]) for nd in L])
        bnd = np.array([len(nd["bnd"]) for nd in L])
        s, b = 3.0 * own, 3.0 * bnd
        # partial factorisation flops (multiply-adds x2): sum_k (m-k)^2 for k < s
        fl = np.array([sum(((si + bi - k) ** 2 for k in range(int(si)))) * 2.0 / 2.0 for si, bi in zip(s, b)])
        tot_flops += fl.sum()
        tot_mem += int(((s + b) ** 2).sum() * 8)
        crit += fl.max()
        print(f"{h:2d} {len(L):6d}   {own.max():4d} / {own.mean():6.1f}   {bnd.max():4d} / {bnd.mean():6.1f}   {int((s + b).max()):5d}"
              f"          {fl.max() / 1e6:9.2f}          {fl.sum() / 1e6:9.1f}    {own.max():4d}")
    print(f"total {tot_flops / 1e6:.0f} Mflop, frontal storage {tot_mem / 1e6:.1f} MB, critical path (largest front per level) {crit / 1e6:.1f} Mflop")
    nnzL = sum(len(nd["own"]) * (len(nd["own"]) + 1) // 2 + len(nd["own"]) * len(nd["bnd"]) for nd in nodes)
    print(f"nnz(L) blocks {nnzL}  ({nnzL * 72 / 1e6:.1f} MB)")


if __name__ == "__main__":
    main()
