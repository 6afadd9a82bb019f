"""Seeded fuzzing of the HIP paths against the oracle (pytest -m gpu): random small graphs with multi-edges,
self-loops, hubs and dead ends; random p/q, walk lengths, batch sizes, walker bases, kernel variants.
Every case is bit-exact or the test fails with the case's parameters in the message."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROUNDS = int(os.environ.get("TRW_FUZZ_ROUNDS", "1"))   # a campaign: TRW_FUZZ_ROUNDS=40 pytest -m gpu tests/test_gpu_fuzz.py
PQS = [(1.0, 1.0), (0.25, 4.0), (4.0, 0.25), (0.7, 0.5), (1.0, 2.0), (3.0, 1.0), (1.0, 1.0 + 1e-9), (0.01, 50.0)]


def _graph(rng, n, with_dead_ends):
    """CSR with rows sorted, multi-edges and self-loops allowed; optionally some empty rows."""
    m = int(rng.integers(n, 6 * n))
    src = rng.integers(0, n, m)
    dst = rng.integers(0, n, m)
    hub = int(rng.integers(0, n))
    k = int(rng.integers(1, max(2, n // 2)))
    spokes = rng.integers(0, n, k)
    src = np.concatenate([src, np.full(k, hub), spokes, rng.integers(0, n, 5).repeat(3)])
    dst = np.concatenate([dst, spokes, np.full(k, hub), rng.integers(0, n, 5).repeat(3)])      # + multi-edges
    if not with_dead_ends:
        src = np.concatenate([src, np.arange(n)])
        dst = np.concatenate([dst, (np.arange(n) + 1) % n])
    else:
        keep = src % 5 != 2
        src, dst = src[keep], dst[keep]
    order = np.lexsort((dst, src))
    src, dst = src[order], dst[order]
    row_ptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(src, minlength=n), out=row_ptr[1:])
    return row_ptr, dst.astype(np.int64), src.astype(np.int64)


@pytest.mark.parametrize("case", range(24 * ROUNDS))
def test_fuzz_csr_walk(case, monkeypatch):
    from torch_rw import rw
    rng = np.random.default_rng(1000 + case)
    n = int(rng.integers(2, 400))
    rp, ci, _ = _graph(rng, n, with_dead_ends=False)
    N = int(rng.integers(1, 3000))
    L = int(rng.integers(1, 45))
    seed = int(rng.integers(-2**31, 2**31))
    base = int(rng.choice([0, 1, 77, 2**32 + 3]))
    p, q = PQS[int(rng.integers(0, len(PQS)))]
    tg = rng.integers(0, n, N)
    env = [{}, {"TRW_HASH": "0"}, {"TRW_HASH": "1", "TRW_EXT": "1"}, {"TRW_EXT": "1"}, {"TRW_FORCE_WIDE": "1"},
           {"TRW_NO_POS": "1"}][case % 6]
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    d = [torch.as_tensor(a).to(DEV) for a in (rp, ci, tg)]
    msg = "case %d n=%d N=%d L=%d seed=%d base=%d p=%g q=%g env=%s" % (case, n, N, L, seed, base, p, q, env)
    if p == 1.0 and q == 1.0:
        ref = O.walk(rp, ci, tg, p, q, L, seed, rng_mode=O.RNG_GLIBC_INDEXED, walker_base=base)
        got = rw.walk_ex(*d, p, q, L, seed, walker_base=base, rng_mode=rw.RNG_GLIBC)
        assert (got.cpu().numpy() == ref).all(), msg
    ref = O.walk(rp, ci, tg, p, q, L, seed, rng_mode=O.RNG_WALKER, walker_base=base)
    got = rw.walk_ex(*d, p, q, L, seed, walker_base=base, rng_mode=rw.RNG_WALKER)
    assert (got.cpu().numpy() == ref).all(), msg
    if N * L <= 20000 and not (p == 1.0 and q == 1.0):
        ref = O.walk(rp, ci, tg, p, q, L, seed, rng_mode=O.RNG_GLIBC_SEQ)
        got = rw.walk_ex(*d, p, q, L, seed, rng_mode=rw.RNG_GLIBC)
        assert (got.cpu().numpy() == ref).all(), msg + " (serial replay)"


@pytest.mark.parametrize("case", range(12 * ROUNDS))
def test_fuzz_edge_list_and_triples(case, monkeypatch):
    from torch_rw import rw, utils
    rng = np.random.default_rng(5000 + case)
    n = int(rng.integers(2, 300))
    _, ci, src = _graph(rng, n, with_dead_ends=True)
    shuffle = rng.permutation(len(src))                      # tails unsorted within a head, like the reference's builder
    edges = torch.as_tensor(np.stack([src[shuffle], ci[shuffle]], 1))
    index, edges_sorted = utils.build_node_edge_index(edges, torch.arange(n))
    N = int(rng.integers(1, 1500))
    L = int(rng.integers(1, 30))
    seed = int(rng.integers(0, 2**31))
    pad = n if case % 3 else int(rng.integers(0, n))
    p, q = PQS[int(rng.integers(0, len(PQS)))]
    restart = bool(case % 2)
    tg = rng.integers(0, n, N)
    tg[::11] = pad
    if case % 4 == 1:
        monkeypatch.setenv("TRW_HASH", "0")
    d = [edges_sorted.to(DEV), index.to(DEV), torch.as_tensor(tg).to(DEV)]
    msg = "case %d n=%d N=%d L=%d seed=%d pad=%d p=%g q=%g restart=%s" % (case, n, N, L, seed, pad, p, q, restart)
    ref = O.walk_edge_list(edges_sorted.numpy(), index.numpy(), tg, p, q, L, seed, pad, restart=restart,
                           rng_mode=O.RNG_WALKER, walker_base=5)
    got = rw.walk_edge_list_ex(*d, p, q, L, seed, pad, restart=restart, walker_base=5, rng_mode=rw.RNG_WALKER)
    assert (got.cpu().numpy() == ref).all(), msg
    if N * L <= 20000:
        ref = O.walk_edge_list(edges_sorted.numpy(), index.numpy(), tg, p, q, L, seed, pad, restart=restart,
                               rng_mode=O.RNG_GLIBC_SEQ)
        got = rw.walk_edge_list_ex(*d, p, q, L, seed, pad, restart=restart, rng_mode=rw.RNG_GLIBC)
        assert (got.cpu().numpy() == ref).all(), msg + " (serial replay)"
    # the same edges as a knowledge graph with random relations
    rel = rng.integers(n, n + 7, len(src))
    triples = torch.as_tensor(np.stack([src[shuffle], rel, ci[shuffle]], 1))
    tindex, tsorted = utils.build_relation_tail_index(triples, torch.arange(n))
    kpad = n + 7
    tgk = np.where(tg == pad, kpad, tg)
    ref = O.walk_triples(tsorted.numpy(), tindex.numpy(), tgk, L, kpad, seed, rng_mode=O.RNG_WALKER, walker_base=9)
    got = rw.walk_triples_ex(tsorted.to(DEV), tindex.to(DEV), torch.as_tensor(tgk).to(DEV), L, kpad, seed,
                             walker_base=9, rng_mode=rw.RNG_WALKER)
    assert (got.cpu().numpy() == ref).all(), msg + " (triples)"


@pytest.mark.parametrize("case", range(8 * ROUNDS))
def test_fuzz_windows(case):
    """All four window builders on random shapes, against the oracle (glibc stream for the fixed-count
    negatives, both modes for the redrawn ones; shards by walk/window base)."""
    from torch_rw import rw
    rng = np.random.default_rng(9000 + case)
    cols = int(rng.integers(1, 60))
    ws = int(rng.integers(1, cols + 1))
    num_walks = int(rng.integers(1, 400))
    num_nodes = int(rng.choice([1, 2, 7, 1000, 10**6, 2**33]))
    seed = int(rng.integers(-2**31, 2**31))
    walks = rng.integers(0, min(num_nodes, 10**6), (num_walks, cols))
    d_w = torch.as_tensor(walks).to(DEV)
    msg = "case %d cols=%d ws=%d walks=%d nodes=%d seed=%d" % (case, cols, ws, num_walks, num_nodes, seed)
    t, p, n = O.to_windows(walks, ws, num_nodes, seed)
    gt, gp, gn = rw.to_windows(d_w, ws, num_nodes, seed)
    assert (gt.cpu().numpy() == t).all() and (gp.cpu().numpy() == p).all() and (gn.cpu().numpy() == n).all(), msg
    cut = int(rng.integers(0, num_walks))
    steps = cols - ws + 1
    st, sp, sn = rw.to_windows_ex(d_w[cut:], ws, num_nodes, seed, walk_base=cut)
    assert (sn.cpu().numpy() == n[cut * steps:]).all() and (sp.cpu().numpy() == p[cut * steps:]).all(), msg + " (shard)"
    if ws >= 2:
        pn, nn, w = O.to_windows_cbow(walks, ws, num_nodes, seed, rng_mode=O.RNG_WALKER, window_base=5)
        gpn, gnn, gw = rw.to_windows_cbow_ex(d_w, ws, num_nodes, seed, window_base=5, rng_mode=rw.RNG_WALKER)
        assert (gpn.cpu().numpy() == pn).all() and (gnn.cpu().numpy() == nn).all() and (gw.cpu().numpy() == w).all(), msg + " (cbow)"
    # triples windows over the same matrix read as a triples walk (odd width), random pool
    tcols = cols if cols % 2 == 1 else cols - 1
    if tcols >= 3:
        tws = int(rng.integers(1, 6))
        pool = rng.integers(0, 50, (int(rng.integers(1, 40)), 3))
        tw = walks[:, :tcols] % 50
        tt, tp, tn = O.to_windows_triples(tw, tws, 50, -1, pool, seed)
        g = rw.to_windows_triples(torch.as_tensor(tw).to(DEV), tws, 50, -1, torch.as_tensor(pool).to(DEV), seed)
        assert (g[0].cpu().numpy() == tt).all() and (g[1].cpu().numpy() == tp).all() and (g[2].cpu().numpy() == tn).all(), \
            msg + " (triples tws=%d pool=%d)" % (tws, len(pool))
