"""Edge-list walk: oracle pinned on the reference's six CPU goldens, host helpers,
and (-m gpu) HIP parity.  Reference: csrc/cpu/rw_cpu_edge_list.cpp, tests/test_rw_edge_list.py."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"


@pytest.fixture(scope="module")
def egold():
    with open(os.path.join(ROOT, "tests", "golden", "reference_test_rw_edge_list.json")) as f:
        return json.load(f)


def _reference_inputs(egold, directed):
    """What the reference's tests build before calling walk_edge_list (tests/test_rw_edge_list.py:13-40)."""
    import networkx as nx
    from torch_rw import utils
    g = nx.DiGraph() if directed else nx.Graph()
    for a, b in egold["edges"]:
        g.add_edge(a, b)
    edge_list_indexed, node_idx_map = utils.to_edge_list_indexed(g)
    target_nodes = torch.Tensor(list(node_idx_map.values())).to(int).contiguous()
    nodes_tensor = torch.unique(edge_list_indexed.view(-1))
    node_edge_index, edges_sorted = utils.build_node_edge_index(edge_list_indexed, nodes_tensor)
    padding_idx = sorted(target_nodes.tolist())[-1] + 1
    return edges_sorted, node_edge_index, target_nodes, padding_idx


def _random_edge_list(n, m, seed, dead_ends=True):
    """Directed edge list with some nodes that have no out-edges; sorted by head, tails in input order."""
    from torch_rw import utils
    rng = np.random.default_rng(seed)
    src = rng.integers(0, n, m)
    if dead_ends:
        src = src[src % 7 != 3]                      # every node = 3 mod 7 is a dead end
    dst = rng.integers(0, n, src.size)
    edges = torch.as_tensor(np.stack([src, dst], 1), dtype=torch.int64)
    index, edges_sorted = utils.build_node_edge_index(edges, torch.arange(n))
    return edges_sorted, index


# ---- CPU tier ------------------------------------------------------------------------------

def test_host_helpers_match_reference_fixtures(egold):
    for directed, key in ((True, "directed"), (False, "undirected")):
        edges, index, targets, pad = _reference_inputs(egold, directed)
        assert index.tolist() == egold["node_edge_index"][key]["rows"]
        assert pad == 5 and edges.dtype == torch.int64
        assert (edges[1:, 0] >= edges[:-1, 0]).all()
    assert _reference_inputs(egold, False)[2].tolist() == [0, 1, 2, 4, 3]   # dict order of the undirected walk


@pytest.mark.parametrize("mode", [O.RNG_GLIBC_SEQ, O.RNG_LIBC])
def test_oracle_reproduces_all_reference_goldens(egold, mode):
    for c in egold["cases"]:
        edges, index, targets, pad = _reference_inputs(egold, c["directed"])
        w = O.walk_edge_list(edges.numpy(), index.numpy(), targets.numpy(), c["p"], c["q"], 6, c["seed"], pad,
                             restart=c["restart"], rng_mode=mode)
        assert w.tolist() == c["walks"], c["source"]


def test_oracle_semantics():
    edges, index = _random_edge_list(120, 500, 3)
    tg = np.random.default_rng(0).integers(0, 120, 400)
    pad = 120
    for p, q, restart in ((1.0, 1.0, True), (1.0, 1.0, False), (0.5, 2.0, True), (2.0, 0.5, False)):
        a, da = O.walk_edge_list(edges.numpy(), index.numpy(), tg, p, q, 15, 7, pad, restart=restart,
                                 rng_mode=O.RNG_GLIBC_SEQ, return_draws=True)
        b, db = O.walk_edge_list(edges.numpy(), index.numpy(), tg, p, q, 15, 7, pad, restart=restart,
                                 rng_mode=O.RNG_LIBC, return_draws=True)
        assert (a == b).all() and da == db
        assert (a[:, 0] == tg).all()
        if not restart:                                            # padding is absorbing without restart
            padded = a == pad
            assert (padded[:, 1:] >= padded[:, :-1]).all()
        else:                                                      # with restart the walk re-enters at its start
            after_pad = a[:, 1:][a[:, :-1] == pad]
            starts = np.repeat(a[:, :1], a.shape[1] - 1, 1)[a[:, :-1] == pad]
            assert (after_pad == starts).all() if (p == 1.0 and q == 1.0) else True
        full = O.walk_edge_list(edges.numpy(), index.numpy(), tg, p, q, 15, 7, pad, restart=restart,
                                rng_mode=O.RNG_WALKER)
        parts = [O.walk_edge_list(edges.numpy(), index.numpy(), tg[lo:hi], p, q, 15, 7, pad, restart=restart,
                                  rng_mode=O.RNG_WALKER, walker_base=lo) for lo, hi in ((0, 9), (9, 400))]
        assert (np.concatenate(parts) == full).all()


def test_surface_names():
    from torch_rw import rw, utils
    import inspect
    import torch_rw_native
    assert list(inspect.signature(rw.walk_edge_list).parameters) == \
        ["edge_list_indexed", "node_edge_index", "target_nodes", "p", "q", "walk_length", "seed", "padding_idx",
         "restart"]                                                  # torch_rw/rw.py:6
    assert callable(torch_rw_native.walk_edge_list) and callable(utils.to_edge_list_indexed)


def test_host_leg_all_reference_goldens_through_the_reference_call(egold):
    """The reference's six CPU goldens (tests/test_rw_edge_list.py) through its own call on CPU tensors: the
    dispatcher's host leg (rw_init.cpp:41-44 -> walk_edge_list_cpu), product code in csrc/trw_cpu.cpp."""
    from torch_rw import rw
    for c in egold["cases"]:
        edges, index, targets, pad = _reference_inputs(egold, c["directed"])
        kw = dict(edge_list_indexed=edges, node_edge_index=index, target_nodes=targets, p=c["p"], q=c["q"],
                  walk_length=6, seed=c["seed"], padding_idx=pad)
        if not c["restart"]:
            kw["restart"] = False
        walks = rw.walk_edge_list(**kw)
        assert walks.device.type == "cpu" and walks.dtype == torch.int64 and tuple(walks.shape) == (5, 7)
        assert walks.tolist() == c["walks"], c["source"]


@pytest.mark.parametrize("p,q", [(1.0, 1.0), (0.25, 4.0), (0.7, 0.2), (4.0, 0.25)])
@pytest.mark.parametrize("restart", [True, False])
def test_host_leg_equals_oracle(p, q, restart):
    from torch_rw import rw
    edges, index = _random_edge_list(300, 2500, 11)
    pad = 300
    tg = np.random.default_rng(1).integers(0, 300, 700)
    tg[::23] = pad
    for L in (1, 2, 25):
        ref = O.walk_edge_list(edges.numpy(), index.numpy(), tg, p, q, L, 5, pad, restart=restart, rng_mode=O.RNG_GLIBC_SEQ)
        got = rw.walk_edge_list(edges, index, torch.as_tensor(tg), p, q, L, 5, pad, restart)
        assert (got.numpy() == ref).all()
    with pytest.raises(RuntimeError, match="outside"):
        rw.walk_edge_list(edges, index, torch.tensor([301]), 1.0, 1.0, 4, 1, pad)
    with pytest.raises(RuntimeError, match="expected scalar type Long"):
        rw.walk_edge_list(edges, index.to(torch.int32), torch.tensor([1]), 1.0, 1.0, 4, 1, pad)


# ---- GPU tier ------------------------------------------------------------------------------

@pytest.mark.gpu
def test_gpu_all_reference_goldens(egold):
    """The reference's six CPU goldens through the reference's own call, on the HIP path."""
    from torch_rw import rw
    for c in egold["cases"]:
        edges, index, targets, pad = _reference_inputs(egold, c["directed"])
        kw = dict(edge_list_indexed=edges.to(DEV), node_edge_index=index.to(DEV), target_nodes=targets.to(DEV),
                  p=c["p"], q=c["q"], walk_length=6, seed=c["seed"], padding_idx=pad)
        if not c["restart"]:
            kw["restart"] = False
        walks = rw.walk_edge_list(**kw)
        assert walks.dtype == torch.int64 and tuple(walks.shape) == (5, 7)
        assert walks.cpu().tolist() == c["walks"], c["source"]


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{}, {"TRW_FORCE_WIDE": "1"}, {"TRW_HASH": "0"}, {"TRW_HASH": "1"}, {"TRW_ELPOOL": "0"},
                                 {"TRW_ELPOOL": "1", "TRW_ELPOOL_BLOCKS": "1"}])
@pytest.mark.parametrize("p,q", [(1.0, 1.0), (0.25, 4.0), (0.7, 0.2), (4.0, 0.25)])
@pytest.mark.parametrize("restart", [True, False])
def test_gpu_walker_stream_equals_oracle(env, p, q, restart, monkeypatch):
    from torch_rw import rw
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    edges, index = _random_edge_list(800, 5000, 11)
    pad = 800
    tg = np.random.default_rng(1).integers(0, 800, 3000)
    tg[::23] = pad                                               # start on the padding id
    d = [edges.to(DEV), index.to(DEV), torch.as_tensor(tg).to(DEV)]
    for L in (1, 2, 3, 40):
        ref = O.walk_edge_list(edges.numpy(), index.numpy(), tg, p, q, L, 5, pad, restart=restart,
                               rng_mode=O.RNG_WALKER, walker_base=77)
        got = rw.walk_edge_list_ex(*d, p, q, L, 5, pad, restart=restart, walker_base=77, rng_mode=rw.RNG_WALKER)
        assert (got.cpu().numpy() == ref).all()


@pytest.mark.gpu
@pytest.mark.parametrize("pool", ["0", "1"])
def test_gpu_long_ranges_pad_tails_and_single_edges(pool, monkeypatch):
    """The prepared entries' corner cases: ranges beyond 128 edges (filter masks: candidates are looked at), tails
    that ARE the padding id, heads with exactly one edge (no candidate draw), multi-edges, self loops -- with and
    without restart, bit for bit the oracle's walks."""
    from torch_rw import rw, utils
    monkeypatch.setenv("TRW_ELPOOL", pool)
    rng = np.random.default_rng(5)
    n, pad = 400, 400
    src = np.concatenate([rng.integers(0, 20, 6000),            # 20 hubs with ~300 edges each
                          rng.integers(20, n, 1500),             # sparse rest: many single-edge heads and dead ends
                          np.arange(30, 60), np.arange(60, 90)])
    dst = np.concatenate([rng.integers(0, n, 6000), rng.integers(0, n, 1500),
                          np.full(30, pad),                      # tails that are the padding id
                          np.arange(60, 90)])                    # self loops
    edges = torch.as_tensor(np.stack([src, dst], 1), dtype=torch.int64)
    index, edges_sorted = utils.build_node_edge_index(edges, torch.arange(n))
    tg = rng.integers(0, n, 4000)
    tg[::31] = pad
    d = [edges_sorted.to(DEV), index.to(DEV), torch.as_tensor(tg).to(DEV)]
    for p, q in ((0.25, 4.0), (4.0, 0.25), (0.7, 0.2)):
        for restart in (True, False):
            ref = O.walk_edge_list(edges_sorted.numpy(), index.numpy(), tg, p, q, 30, 5, pad, restart=restart,
                                   rng_mode=O.RNG_WALKER, walker_base=11)
            got = rw.walk_edge_list_ex(*d, p, q, 30, 5, pad, restart=restart, walker_base=11, rng_mode=rw.RNG_WALKER)
            assert (got.cpu().numpy() == ref).all(), (p, q, restart)


@pytest.mark.gpu
def test_gpu_padding_inside_the_index():
    """padding_idx that is itself a row of node_edge_index: is_neighbor(x, padding) reads that row."""
    from torch_rw import rw
    edges, index = _random_edge_list(300, 2000, 13)
    pad = 3                                                       # a dead-end node id (3 mod 7) used as padding
    tg = np.random.default_rng(2).integers(0, 300, 2000)
    d = [edges.to(DEV), index.to(DEV), torch.as_tensor(tg).to(DEV)]
    for p, q in ((1.0, 1.0), (0.5, 2.0)):
        ref = O.walk_edge_list(edges.numpy(), index.numpy(), tg, p, q, 20, 9, pad, rng_mode=O.RNG_WALKER)
        assert (rw.walk_edge_list_ex(*d, p, q, 20, 9, pad, rng_mode=rw.RNG_WALKER).cpu().numpy() == ref).all()


@pytest.mark.gpu
def test_gpu_serial_replay_auto_rule_shards_and_errors():
    from torch_rw import rw
    edges, index = _random_edge_list(400, 2500, 17)
    pad = 400
    tg = np.random.default_rng(3).integers(0, 400, 500)
    d_e, d_i, d_t = edges.to(DEV), index.to(DEV), torch.as_tensor(tg).to(DEV)
    for p, q, n in ((1.0, 1.0, 500), (0.7, 0.2, 404)):          # 404 = 32768 // 81, the biased grain at L=80
        ref = O.walk_edge_list(edges.numpy(), index.numpy(), tg[:n], p, q, 80, 10, pad, rng_mode=O.RNG_GLIBC_SEQ)
        assert (rw.walk_edge_list_ex(d_e, d_i, d_t[:n], p, q, 80, 10, pad, rng_mode=rw.RNG_GLIBC).cpu().numpy()
                == ref).all()
        assert (rw.walk_edge_list(d_e, d_i, d_t[:n], p, q, 80, 10, pad).cpu().numpy() == ref).all()   # AUTO
    big = rw.walk_edge_list(d_e, d_i, d_t, 0.7, 0.2, 80, 10, pad).cpu().numpy()     # 500 > grain: walker stream
    assert (big == O.walk_edge_list(edges.numpy(), index.numpy(), tg, 0.7, 0.2, 80, 10, pad,
                                    rng_mode=O.RNG_WALKER)).all()
    full = rw.walk_edge_list_ex(d_e, d_i, d_t, 0.5, 2.0, 30, 4, pad, rng_mode=rw.RNG_WALKER)
    parts = [rw.walk_edge_list_ex(d_e, d_i, d_t[lo:hi], 0.5, 2.0, 30, 4, pad, walker_base=lo, rng_mode=rw.RNG_WALKER)
             for lo, hi in ((0, 1), (1, 257), (257, 500))]
    assert torch.equal(torch.cat(parts), full)
    assert tuple(rw.walk_edge_list(d_e, d_i, d_t[:0], 1.0, 1.0, 4, 1, pad).shape) == (0, 5)
    with pytest.raises(RuntimeError, match="expected scalar type Long"):
        rw.walk_edge_list(d_e.to(torch.int32), d_i, d_t, 1.0, 1.0, 4, 1, pad)
    with pytest.raises(RuntimeError, match="outside"):
        rw.walk_edge_list_ex(d_e, d_i, torch.tensor([401], device=DEV), 1.0, 1.0, 4, 1, pad, check=True)
    rw.walk_edge_list(d_e, d_i, torch.tensor([401], device=DEV), 1.0, 1.0, 4, 1, pad)     # default: reported later
    with pytest.raises(RuntimeError, match="outside"):
        rw.check_status()


@pytest.mark.gpu
def test_gpu_calls_replay_from_a_hip_graph():
    """Edge-list (table path), triples and window-builder calls captured into one HIP graph and replayed:
    everything a call clears it clears with its own kernels (see test_gpu_parity.test_walk_is_graph_capturable)."""
    from torch_rw import rw
    from tests.test_triples import _kg
    edges, index = _random_edge_list(800, 5000, 11)
    tg = np.random.default_rng(1).integers(0, 800, 6000)
    d = [edges.to(DEV), index.to(DEV), torch.as_tensor(tg).to(DEV)]
    ref_e = O.walk_edge_list(edges.numpy(), index.numpy(), tg, 0.25, 4.0, 30, 5, 800, restart=True, rng_mode=O.RNG_WALKER)
    ts, tindex, ents, pad = _kg(2000, 9000, 11, 9)
    ttg = np.random.default_rng(2).integers(0, 2000, 5000)
    dt = [ts.to(DEV), tindex.to(DEV), torch.as_tensor(ttg).to(DEV)]
    ref_t = O.walk_triples(ts.numpy(), tindex.numpy(), ttg, 20, pad, 3, rng_mode=O.RNG_WALKER)
    walks = np.random.default_rng(3).integers(0, 500, (900, 31))
    d_walks = torch.as_tensor(walks).to(DEV)
    ref_w = O.to_windows(walks, 5, 500, 7)

    def calls():
        return (rw.walk_edge_list_ex(*d, 0.25, 4.0, 30, 5, 800, restart=True, rng_mode=rw.RNG_WALKER, check=False),
                rw.walk_triples_ex(*dt, 20, pad, 3, rng_mode=rw.RNG_WALKER, check=False),
                rw.to_windows(d_walks, 5, 500, 7))

    calls()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        calls()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        we, wt, (t, p, n) = calls()
    for _ in range(3):
        for x in (we, wt, t, p, n):
            x.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert (we.cpu().numpy() == ref_e).all() and (wt.cpu().numpy() == ref_t).all()
        assert (t.cpu().numpy() == ref_w[0]).all() and (p.cpu().numpy() == ref_w[1]).all() and (n.cpu().numpy() == ref_w[2]).all()
