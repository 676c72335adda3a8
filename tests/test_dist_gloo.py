"""N > 1 path on CPU: world_size-2 gloo processes emulate the multi-GPU scheme of libsgo
(DESIGN.md section 6): every rank holds the full graph, evaluates the level-0 Hessian product only for the
rows of its range of tiles (zeros elsewhere), and an all-reduce(sum) of the product vectors must reproduce
the full product exactly (one non-zero contributor per row).  The partition is libsgo's own: the host-only
row plan sgo_plan_rows (Hilbert row order, tiles, tile range per rank -- the code sgo_set_graph_se2 runs) and
sgo_shard_range; the arithmetic of the rows comes from the numpy oracle (no GPU here)."""
import os
import socket

import numpy as np
import pytest

from sparse_gslam_amd import capi, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_range_is_a_balanced_partition():
    for count in (0, 1, 7, 64, 1000, 36959):
        for n in range(1, 9):
            cuts = [capi.shard_range(count, n, r) for r in range(n)]
            assert cuts[0][0] == 0 and cuts[-1][1] == count
            for (a0, a1), (b0, b1) in zip(cuts, cuts[1:]):
                assert a1 == b0 and a0 <= a1
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1


def _rows_product(g, x, vertex_rows):
    """(H x) restricted to the Hessian rows of the vertices `vertex_rows`, zeros elsewhere; H from the numpy
    oracle, x and the result in g2o's hessian order (free vertices in ascending id)."""
    from oracle import np_oracle as no
    hidx, free = no.hessian_index(g.fixed)
    n = free.size
    xi, xj = g.poses[g.ei], g.poses[g.ej]
    e = no.edge_error(xi, xj, g.meas)
    A, B = no.edge_jacobians(xi, xj, g.meas)
    O = no.info_full(g.info)
    e2 = np.einsum("ni,nij,nj->n", e, O, e)
    _, rho1 = no.dcs_rho(e2, g.phi)
    Ow = O * rho1[:, None, None]
    mine = np.zeros(g.poses.shape[0], dtype=bool)
    mine[vertex_rows] = True
    y = np.zeros((n, 3))
    hi, hj = hidx[g.ei], hidx[g.ej]
    At, Bt = np.swapaxes(A, 1, 2), np.swapaxes(B, 1, 2)
    for Jr_t, Jc, hr, hc, vr in ((At, A, hi, hi, g.ei), (Bt, B, hj, hj, g.ej), (At, B, hi, hj, g.ei), (Bt, A, hj, hi, g.ej)):
        m = (hr >= 0) & (hc >= 0) & mine[vr]
        blk = Jr_t[m] @ Ow[m] @ Jc[m]
        np.add.at(y, hr[m], np.einsum("nij,nj->ni", blk, x[hc[m]]))
    return y


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # rendezvous plumbing used by bench.py: rank 0 creates an id, everyone receives it
        uid = [os.urandom(capi.UNIQUE_ID_BYTES) if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        assert isinstance(uid[0], bytes) and len(uid[0]) == capi.UNIQUE_ID_BYTES
        g = synth.manhattan(3000, 12000, seed=21, info_mode="full")      # same graph on every rank
        n = int((~g.fixed).sum())
        # libsgo's own host-side plan (no GPU): which rows of a level-0 product this rank contributes
        plan = capi.plan_rows(g.poses, g.fixed, g.ei, g.ej, world)
        rb = plan["rank_row_begin"]
        ok = plan["n"] == n and rb[0] == 0 and rb[-1] == n and all(rb[k] < rb[k + 1] for k in range(world))
        ok = ok and sorted(plan["row_vertex"].tolist()) == np.flatnonzero(~g.fixed).tolist()
        tb = plan["tile_row_begin"]
        ok = ok and all(int(b) in set(tb.tolist()) for b in rb)           # rank boundaries are tile boundaries
        x = np.random.default_rng(5).standard_normal((n, 3))
        mine = plan["row_vertex"][rb[rank]:rb[rank + 1]]
        y = _rows_product(g, x, mine)
        ty = torch.from_numpy(y.copy())
        dist.all_reduce(ty)                                                # the per-product exchange of the scheme
        full = _rows_product(g, x, np.flatnonzero(~g.fixed))
        ok = ok and bool(np.array_equal(ty.numpy(), full))                 # one contributor per row: exact
        # chi2: per-rank partial sums over an edge range, all-reduced (rounding-level agreement)
        from oracle import np_oracle as no
        e0, e1 = capi.shard_range(g.E, world, rank)
        c, rc, _ = no.chi2(g.poses, g.ei[e0:e1], g.ej[e0:e1], g.meas[e0:e1], g.info[e0:e1], g.phi[e0:e1])
        t = torch.tensor([c, rc], dtype=torch.float64)
        dist.all_reduce(t)
        fc, frc, _ = no.chi2(g.poses, g.ei, g.ej, g.meas, g.info, g.phi)
        ok = ok and abs(t[0].item() - fc) <= 1e-12 * fc and abs(t[1].item() - frc) <= 1e-12 * frc
        q.put((rank, bool(ok), uid[0].hex()))
    finally:
        dist.destroy_process_group()


def test_two_rank_row_partition_reproduces_the_full_product():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert res[0][2] == res[1][2]     # both ranks saw the same rendezvous id


def test_plan_rows_is_a_partition_for_every_world_size():
    g = synth.manhattan(2000, 9000, seed=4)
    for world in (1, 2, 3, 8):
        plan = capi.plan_rows(g.poses, g.fixed, g.ei, g.ej, world)
        rb = plan["rank_row_begin"]
        assert rb[0] == 0 and rb[-1] == plan["n"] and np.all(np.diff(rb) >= 0)
        assert np.array_equal(np.sort(plan["row_vertex"]), np.flatnonzero(~g.fixed))


def test_every_rank_gets_its_own_256_tiles_and_thin_boundaries():
    """A world of G ranks cuts 256 G tiles (each rank keeps its own CUs busy; a one-GPU graph of this size would take the
    wave-group kernel and no tiles at all), rank boundaries sit on tile boundaries, the ranks' shares of the stored blocks
    are even, and -- the premise of the row-owner mode -- only a few per cent of the rows have an edge into another rank's
    range when the closures are spatially local (Hilbert row order)."""
    g = synth.manhattan(40000, 300000, seed=12)
    ntiles = {}
    for world in (1, 2, 4, 8):
        plan = capi.plan_rows(g.poses, g.fixed, g.ei, g.ej, world)
        tb, rb, n = plan["tile_row_begin"], plan["rank_row_begin"], plan["n"]
        ntiles[world] = tb.size - 1
        assert set(rb.tolist()) <= set(tb.tolist())
        if world == 1:
            continue
        row_of = np.full(g.V, -1, dtype=np.int64)
        row_of[plan["row_vertex"]] = np.arange(n)
        ri, rj = row_of[g.ei], row_of[g.ej]
        ok = (ri >= 0) & (rj >= 0)
        qi = np.searchsorted(rb, ri[ok], side="right") - 1
        qj = np.searchsorted(rb, rj[ok], side="right") - 1
        cross = qi != qj
        isb = np.zeros(n, dtype=bool)
        isb[ri[ok][cross]] = True
        isb[rj[ok][cross]] = True
        assert isb.sum() <= 0.25 * n, (world, int(isb.sum()))          # the library's row-owner criterion
        assert isb.sum() <= 0.12 * n, (world, int(isb.sum()))          # in fact a few per cent
        # even shares: the tiles are cut to equal stored blocks, a rank owns ntiles / world of them
        deg = np.bincount(np.concatenate([ri[ok], rj[ok]]), minlength=n)
        share = np.array([deg[rb[q]:rb[q + 1]].sum() for q in range(world)], dtype=float)
        assert share.max() <= 1.25 * share.mean(), (world, share)
    # 256 per rank until a tile would fall below 512 stored blocks (a 300 k-edge graph: ~800 tiles at most)
    assert 240 <= ntiles[1] <= 256 + 8 and ntiles[2] > 1.8 * ntiles[1] and ntiles[8] >= ntiles[4] > 1.5 * ntiles[2]


def test_plan_rows_on_a_large_graph_with_long_range_edges(monkeypatch):
    """E >= 200 000: the slot placement runs as a parallel stable counting sort over chunks of the edge list; 20 %
    random closures and a small LDS budget (the experiment knob SGO_TILE_LDS stands in for the graphs of C5's kind,
    which a CPU test cannot afford): the tile cut is limited by the LDS, many more tiles than CUs, and is taken as it
    is.  The plan must still be a partition of the free vertices into consecutive tiles, with rank boundaries on
    tile boundaries, and must not depend on the call (two calls agree)."""
    monkeypatch.setenv("SGO_TILE_LDS", "6000")
    g = synth.manhattan(30000, 250000, seed=8, p_random=0.2)
    a = capi.plan_rows(g.poses, g.fixed, g.ei, g.ej, 8)
    b = capi.plan_rows(g.poses, g.fixed, g.ei, g.ej, 8)
    assert a["n"] == int((~g.fixed).sum())
    assert np.array_equal(np.sort(a["row_vertex"]), np.flatnonzero(~g.fixed))
    tb, rb = a["tile_row_begin"], a["rank_row_begin"]
    assert tb[0] == 0 and tb[-1] == a["n"] and np.all(np.diff(tb) > 0)
    assert tb.size - 1 >= 4 * 256       # closed by the LDS budget, not by a CU's share of the blocks
    assert rb[0] == 0 and rb[-1] == a["n"] and set(rb.tolist()) <= set(tb.tolist())
    for k in ("row_vertex", "tile_row_begin", "rank_row_begin"):
        assert np.array_equal(a[k], b[k])


def test_row_order_of_a_graph_whose_poses_contradict_its_closures():
    """Host-only (sgo_plan_rows with the measurements, ABI 0.1.6): from a dead-reckoned start the Hilbert order of the POSES puts
    the endpoints of the closures thousands of rows apart; the plan then orders the rows by spanning-tree positions (breadth-first
    over all edges from the fixed vertex) -- a permutation of the same rows, with the closures' endpoints close again -- and
    leaves a consistent graph's order alone."""
    from sparse_gslam_amd import capi, synth
    g = synth.manhattan(20000, 100000, seed=3, init="odom")
    a = capi.plan_rows(g.poses, g.fixed, g.ei, g.ej)
    b = capi.plan_rows(g.poses, g.fixed, g.ei, g.ej, meas=g.meas)
    assert a["n"] == b["n"] and np.array_equal(np.sort(a["row_vertex"]), np.sort(b["row_vertex"]))

    def spread(plan):
        hp = np.full(g.V, -1)
        hp[plan["row_vertex"]] = np.arange(plan["n"])
        m = (hp[g.ei] >= 0) & (hp[g.ej] >= 0)
        return float(np.median(np.abs(hp[g.ei][m] - hp[g.ej][m])))
    assert spread(b) < 0.25 * spread(a), (spread(a), spread(b))
    c = synth.manhattan(20000, 100000, seed=3)          # near the optimum: the poses agree with the closures
    pa = capi.plan_rows(c.poses, c.fixed, c.ei, c.ej)
    pb = capi.plan_rows(c.poses, c.fixed, c.ei, c.ej, meas=c.meas)
    assert np.array_equal(pa["row_vertex"], pb["row_vertex"])
