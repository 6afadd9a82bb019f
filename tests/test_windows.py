"""Window builders (first half: to_windows, to_windows_cbow): oracle pinned on the
reference's goldens, and (-m gpu) HIP parity.  Reference: csrc/cpu/windows_cpu.cpp, tests/test_windows.py."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"


@pytest.fixture(scope="module")
def wgold():
    with open(os.path.join(ROOT, "tests", "golden", "reference_test_windows.json")) as f:
        return json.load(f)


# ---- CPU tier ------------------------------------------------------------------------------

@pytest.mark.parametrize("mode", [O.RNG_GLIBC_SEQ, O.RNG_LIBC])
def test_oracle_reproduces_reference_goldens(wgold, mode):
    g = wgold
    t, p, n = O.to_windows(g["walks"], g["window_size"], g["num_nodes"], g["seed"], rng_mode=mode)
    e = g["to_windows_cpu"]
    assert t.shape == (18,) and p.shape == (18, 4) and n.shape == (18, 4)
    assert t[:6].tolist() == e["target_nodes"] and p[:6].tolist() == e["pos_windows"]
    assert n[:6].tolist() == e["neg_windows"]
    pn, nn, w = O.to_windows_cbow(g["walks"], g["window_size"], g["num_nodes"], g["seed"], rng_mode=mode)
    e = g["to_windows_cbow_cpu"]
    assert pn[:6].tolist() == e["pos_nodes"] and nn[:6].tolist() == e["neg_nodes"] and w[:6].tolist() == e["windows"]


def test_fixture_walks_are_what_the_reference_test_draws(wgold):
    torch.manual_seed(20)
    assert torch.randint(low=0, high=30, size=(3, 10)).tolist() == wgold["walks"]   # tests/test_windows.py:6-7


def test_oracle_window_semantics():
    rng = np.random.default_rng(0)
    walks = rng.integers(0, 50, (40, 21))
    for ws in (1, 2, 5, 8, 21):
        t, p, n = O.to_windows(walks, ws, 50, 3)
        steps = 21 - ws + 1
        assert t.shape == (40 * steps,) and p.shape == (40 * steps, ws - 1)
        mid = ws // 2
        st = min(2, steps - 1)
        w = 7 * steps + st                                    # walk 7, step st
        assert t[w] == walks[7, st + mid]
        assert p[w].tolist() == [walks[7, st + i] for i in range(ws) if i != mid]
        assert (n.reshape(-1) == O.glibc_stream(3, n.size) % 50).all()     # the negatives ARE the stream
        pn, nn, cw = O.to_windows_cbow(walks, ws, 50, 3)
        assert (pn == t).all() and (cw == p).all() and (nn != pn).mean() > 0.999  # 101 tries at 1/50 each
    # a collision forces a redraw: with 2 nodes half of the first draws collide
    pn, nn, _ = O.to_windows_cbow(rng.integers(0, 2, (50, 9)), 3, 2, 5)
    assert (nn != pn).all()
    # num_nodes = 1: sample_int draws nothing and the loop gives up after 101 tries
    pn, nn, _ = O.to_windows_cbow(np.zeros((3, 5), dtype=np.int64), 3, 1, 5)
    assert (nn == 0).all()
    with pytest.raises(O.OracleError):
        O.to_windows(walks, 22, 50, 3)


def test_surface_names():
    from torch_rw import rw
    import inspect
    import torch_rw_native
    assert list(inspect.signature(rw.to_windows).parameters) == ["walks", "window_size", "num_nodes", "seed"]
    assert list(inspect.signature(rw.to_windows_cbow).parameters) == ["walks", "window_size", "num_nodes", "seed"]
    assert callable(torch_rw_native.to_windows) and callable(torch_rw_native.to_windows_cbow)


def test_host_legs_reference_goldens_and_oracle(wgold):
    """All four window builders on CPU tensors (rw_init.cpp:77-131 -> to_windows*_cpu): the reference's CPU goldens
    through its own calls, then element for element against the oracle on random shapes."""
    from torch_rw import rw
    g = wgold
    walks = torch.tensor(g["walks"])
    t, p, n = rw.to_windows(walks=walks, window_size=5, num_nodes=30, seed=20)
    e = g["to_windows_cpu"]
    assert t.device.type == "cpu" and t.size(0) == 18
    assert t[:6].tolist() == e["target_nodes"] and p[:6].tolist() == e["pos_windows"] and n[:6].tolist() == e["neg_windows"]
    pn, nn, w = rw.to_windows_cbow(walks=walks, window_size=5, num_nodes=30, seed=20)
    e = g["to_windows_cbow_cpu"]
    assert pn[:6].tolist() == e["pos_nodes"] and nn[:6].tolist() == e["neg_nodes"] and w[:6].tolist() == e["windows"]
    c = g["triples_case"]
    cw, ct = torch.tensor(c["walks"]), torch.tensor(c["triples"])
    tt, pw, nw = rw.to_windows_triples(walks=cw, window_size=4, num_nodes=30, padding_idx=-1, triples=ct, seed=20)
    e = c["to_windows_triples_cpu"]
    assert tt[:2].tolist() == e["target_triples"] and pw[:2].tolist() == e["pos_windows"] and nw[:2].tolist() == e["neg_windows"]
    pt, nt, w2 = rw.to_windows_triples_cbow(walks=cw, window_size=4, num_nodes=30, padding_idx=-1, triples=ct, seed=20)
    e2 = c["to_windows_triples_cbow_cpu"]
    assert pt[:2].tolist() == e2["pos_triples"] and nt[:2].tolist() == e2["neg_triples"] and w2[:2].tolist() == e["pos_windows"]
    rng = np.random.default_rng(3)
    for ws, num_nodes, seed in ((1, 7, 1), (2, 1, 3), (5, 30, 20), (8, 10**6, -1), (11, 2**33, 10)):
        wk = rng.integers(0, min(num_nodes, 1000), (37, 23))
        ref = O.to_windows(wk, ws, num_nodes, seed)
        got = rw.to_windows(torch.as_tensor(wk), ws, num_nodes, seed)
        assert all((a.numpy() == b).all() for a, b in zip(got, ref))
        ref = O.to_windows_cbow(wk, ws, num_nodes, seed)
        got = rw.to_windows_cbow(torch.as_tensor(wk), ws, num_nodes, seed)
        assert all((a.numpy() == b).all() for a, b in zip(got, ref))
    for ws, cols, ntr in ((1, 3, 1), (2, 7, 5), (4, 21, 10), (3, 161, 1000), (5, 20, 3)):
        wk = rng.integers(0, 50, (9, cols))
        tr = rng.integers(0, 50, (ntr, 3))
        ref = O.to_windows_triples(wk, ws, 50, 49, tr, 7)
        got = rw.to_windows_triples(torch.as_tensor(wk), ws, 50, 49, torch.as_tensor(tr), 7)
        assert all((a.numpy() == b).all() for a, b in zip(got, ref))
        ref = O.to_windows_triples_cbow(wk, ws, 50, 49, tr, 7)
        got = rw.to_windows_triples_cbow(torch.as_tensor(wk), ws, 50, 49, torch.as_tensor(tr), 7)
        assert all((a.numpy() == b).all() for a, b in zip(got, ref))


# ---- GPU tier ------------------------------------------------------------------------------

@pytest.mark.gpu
def test_gpu_reference_goldens(wgold):
    from torch_rw import rw
    g = wgold
    walks = torch.tensor(g["walks"], device=DEV)
    target_nodes, pos_windows, neg_windows = rw.to_windows(walks=walks, window_size=5, num_nodes=30, seed=20)
    e = g["to_windows_cpu"]
    assert target_nodes.size(0) == 6 * 3 and target_nodes.dtype == torch.int64
    assert target_nodes[:6].cpu().tolist() == e["target_nodes"]
    assert pos_windows[:6].cpu().tolist() == e["pos_windows"]
    assert neg_windows[:6].cpu().tolist() == e["neg_windows"]          # the reference's CPU golden
    pos_nodes, neg_nodes, windows = rw.to_windows_cbow(walks=walks, window_size=5, num_nodes=30, seed=20)
    e = g["to_windows_cbow_cpu"]
    assert pos_nodes[:6].cpu().tolist() == e["pos_nodes"] and neg_nodes[:6].cpu().tolist() == e["neg_nodes"]
    assert windows[:6].cpu().tolist() == e["windows"]


@pytest.mark.gpu
@pytest.mark.parametrize("ws,num_nodes,seed", [(1, 7, 1), (2, 1, 3), (5, 30, 20), (8, 10**6, -1), (11, 2**33, 10)])
def test_gpu_to_windows_equals_oracle(ws, num_nodes, seed):
    from torch_rw import rw
    walks = np.random.default_rng(ws).integers(0, min(num_nodes, 10**6), (700, 41))
    t, p, n = O.to_windows(walks, ws, num_nodes, seed)
    gt, gp, gn = rw.to_windows(torch.as_tensor(walks).to(DEV), ws, num_nodes, seed)
    assert (gt.cpu().numpy() == t).all() and (gp.cpu().numpy() == p).all() and (gn.cpu().numpy() == n).all()
    # a shard of the walks, told where it starts, produces exactly its rows
    steps = 41 - ws + 1
    st, sp, sn = rw.to_windows_ex(torch.as_tensor(walks[300:]).to(DEV), ws, num_nodes, seed, walk_base=300)
    assert (sn.cpu().numpy() == n[300 * steps:]).all() and (st.cpu().numpy() == t[300 * steps:]).all()


@pytest.mark.gpu
def test_gpu_to_windows_full_size_properties():
    """10 M-walk scale is bench territory; here 200k walks x 81 columns (15.4 M windows, 61 M negatives)."""
    from torch_rw import rw
    g = torch.Generator(device=DEV).manual_seed(5)
    walks = torch.randint(0, 10_000_000, (200_000, 81), generator=g, device=DEV)
    t, p, n = rw.to_windows(walks, 5, 10_000_000, 10)
    assert tuple(n.shape) == (200_000 * 77, 4) and int(n.min()) >= 0 and int(n.max()) < 10_000_000
    flat = n.reshape(-1)
    for k in (0, 1, 123_456, 7_700_000 * 4 + 3, flat.numel() - 1):     # spot draws against the stream
        assert int(flat[k]) == int(O.glibc_stream(10, 1, start=k)[0]) % 10_000_000
    assert torch.equal(t.view(200_000, 77), walks[:, 2:79])
    assert torch.equal(p.view(200_000, 77, 4)[:, :, 0], walks[:, 0:77])
    assert torch.equal(p.view(200_000, 77, 4)[:, :, 3], walks[:, 4:81])


@pytest.mark.gpu
def test_gpu_to_windows_long_runs():
    """Large jobs give every lane a longer run of the stream (fewer jump-aheads): 700k walks x 81 columns =
    215 M negatives; spot draws everywhere, every lane boundary of the first waves, and a shard."""
    from torch_rw import rw
    g = torch.Generator(device=DEV).manual_seed(6)
    walks = torch.randint(0, 999_983, (700_000, 81), generator=g, device=DEV)
    t, p, n = rw.to_windows(walks, 5, 999_983, 77)
    flat = n.reshape(-1)
    total = flat.numel()
    assert total == 700_000 * 77 * 4
    ks = set(np.random.default_rng(3).integers(0, total, 150).tolist())
    ks |= {0, total - 1}
    for run in (124, 155, 186, 217):                      # whichever run length the library chose
        ks |= {run * l + d for l in (1, 2, 63, 64, 65, 255, 256, 257) for d in (-1, 0, 1)}
    for k in sorted(ks):
        assert int(flat[k]) == int(O.glibc_stream(77, 1, start=k)[0]) % 999_983, k
    # 64 consecutive lanes' worth of draws, contiguous
    ref = O.glibc_stream(77, 20_000, start=0) % 999_983
    assert (flat[:20_000].cpu().numpy() == ref).all()
    st, sp, sn = rw.to_windows_ex(walks[650_000:], 5, 999_983, 77, walk_base=650_000)
    assert torch.equal(sn, n[650_000 * 77:]) and torch.equal(sp, p[650_000 * 77:]) and torch.equal(st, t[650_000 * 77:])
    assert torch.equal(t.view(700_000, 77), walks[:, 2:79])
    assert torch.equal(p.view(700_000, 77, 4)[:, :, 1], walks[:, 1:78])


@pytest.mark.gpu
def test_gpu_cbow_modes_equal_oracle():
    from torch_rw import rw
    walks = np.random.default_rng(3).integers(0, 12, (900, 30))         # 12 nodes: many collisions
    d = torch.as_tensor(walks).to(DEV)
    pn, nn, w = O.to_windows_cbow(walks, 4, 12, 9, rng_mode=O.RNG_GLIBC_SEQ)
    gp, gn, gw = rw.to_windows_cbow(d, 4, 12, 9)                        # AUTO: 24300 windows -> serial replay
    assert (gp.cpu().numpy() == pn).all() and (gn.cpu().numpy() == nn).all() and (gw.cpu().numpy() == w).all()
    pn, nn, w = O.to_windows_cbow(walks, 4, 12, 9, rng_mode=O.RNG_WALKER, window_base=1 << 34)
    gp, gn, gw = rw.to_windows_cbow_ex(d, 4, 12, 9, window_base=1 << 34, rng_mode=rw.RNG_WALKER)
    assert (gp.cpu().numpy() == pn).all() and (gn.cpu().numpy() == nn).all() and (gw.cpu().numpy() == w).all()
    one = torch.zeros((4, 6), dtype=torch.int64, device=DEV)
    for mode in (rw.RNG_GLIBC, rw.RNG_WALKER):
        assert (rw.to_windows_cbow_ex(one, 3, 1, 5, rng_mode=mode)[1] == 0).all()
    with pytest.raises(RuntimeError, match="contigous"):
        rw.to_windows(d[:, ::2], 3, 12, 1)
    with pytest.raises(RuntimeError, match="window_size"):
        rw.to_windows(d, 31, 12, 1)
    with pytest.raises(RuntimeError, match="num_nodes"):
        rw.to_windows(d, 3, 0, 1)


# ---- triples window builders (windows_cpu.cpp:160-474) ------------------------------------------

@pytest.mark.parametrize("mode", [O.RNG_GLIBC_SEQ, O.RNG_LIBC])
def test_oracle_reproduces_triples_window_goldens(wgold, mode):
    c = wgold["triples_case"]
    t, p, n = O.to_windows_triples(c["walks"], c["window_size"], c["num_nodes"], c["padding_idx"], c["triples"],
                                   c["seed"], rng_mode=mode)
    e = c["to_windows_triples_cpu"]
    assert t.shape == (30, 3) and p.shape == (30, 8, 3) and n.shape == (30, 8, 3)
    assert t[:2].tolist() == e["target_triples"] and p[:2].tolist() == e["pos_windows"]
    assert n[:2].tolist() == e["neg_windows"]
    pt, nt, w = O.to_windows_triples_cbow(c["walks"], c["window_size"], c["num_nodes"], c["padding_idx"],
                                          c["triples"], c["seed"], rng_mode=mode)
    e2 = c["to_windows_triples_cbow_cpu"]
    assert pt[:2].tolist() == e2["pos_triples"] and nt[:2].tolist() == e2["neg_triples"]
    assert w[:2].tolist() == e["pos_windows"]


def test_triples_fixture_is_what_the_reference_test_draws(wgold):
    torch.manual_seed(20)
    c = wgold["triples_case"]
    assert torch.randint(low=0, high=30, size=(3, 21)).tolist() == c["walks"]        # tests/test_windows.py:124
    assert torch.randint(low=0, high=30, size=(10, 3)).tolist() == c["triples"]      # :125


def test_surface_names_triples():
    from torch_rw import rw
    import inspect
    assert list(inspect.signature(rw.to_windows_triples).parameters) == \
        ["walks", "window_size", "num_nodes", "padding_idx", "triples", "seed"]          # torch_rw/rw.py:35
    assert list(inspect.signature(rw.to_windows_triples_cbow).parameters) == \
        ["walks", "window_size", "num_nodes", "padding_idx", "triples", "seed"]


@pytest.mark.gpu
def test_gpu_triples_window_goldens(wgold):
    from torch_rw import rw
    c = wgold["triples_case"]
    walks, triples = torch.tensor(c["walks"], device=DEV), torch.tensor(c["triples"], device=DEV)
    target_triples, pos_windows, neg_windows = rw.to_windows_triples(walks=walks, window_size=4, num_nodes=30,
                                                                     padding_idx=-1, triples=triples, seed=20)
    e = c["to_windows_triples_cpu"]
    assert target_triples[:2].cpu().tolist() == e["target_triples"]
    assert pos_windows[:2].cpu().tolist() == e["pos_windows"]
    assert neg_windows[:2].cpu().tolist() == e["neg_windows"]           # the reference's CPU golden
    pos_triples, neg_triples, windows = rw.to_windows_triples_cbow(walks=walks, window_size=4, num_nodes=30,
                                                                   padding_idx=-1, triples=triples, seed=20)
    e2 = c["to_windows_triples_cbow_cpu"]
    assert pos_triples[:2].cpu().tolist() == e2["pos_triples"] and neg_triples[:2].cpu().tolist() == e2["neg_triples"]
    assert windows[:2].cpu().tolist() == e["pos_windows"]


@pytest.mark.gpu
@pytest.mark.parametrize("ws,cols,ntr", [(1, 3, 1), (2, 7, 5), (4, 21, 10), (3, 161, 1000), (5, 20, 3)])
def test_gpu_triples_windows_equal_oracle(ws, cols, ntr):
    from torch_rw import rw
    rng = np.random.default_rng(ws * 100 + cols)
    walks = rng.integers(0, 40, (500, cols))
    triples = rng.integers(0, 40, (ntr, 3))
    if ntr >= 3:
        walks[::7, 0:3] = triples[1]                                     # some positives that are in the pool
    d_w, d_t = torch.as_tensor(walks).to(DEV), torch.as_tensor(triples).to(DEV)
    t, p, n = O.to_windows_triples(walks, ws, 40, -7, triples, 11)
    gt, gp, gn = rw.to_windows_triples(d_w, ws, 40, -7, d_t, 11)
    assert (gt.cpu().numpy() == t).all() and (gp.cpu().numpy() == p).all() and (gn.cpu().numpy() == n).all()
    per_walk = (cols - 1) // 2
    st, sp, sn = rw.to_windows_triples_ex(d_w[200:], ws, -7, d_t, 11, base=200)
    assert (sn.cpu().numpy() == n[200 * per_walk:]).all() and (sp.cpu().numpy() == p[200 * per_walk:]).all()
    pt, nt, w = O.to_windows_triples_cbow(walks, ws, 40, -7, triples, 11, rng_mode=O.RNG_GLIBC_SEQ)
    gpt, gnt, gw = rw.to_windows_triples_ex(d_w, ws, -7, d_t, 11, cbow=True, rng_mode=rw.RNG_GLIBC)
    assert (gpt.cpu().numpy() == pt).all() and (gnt.cpu().numpy() == nt).all() and (gw.cpu().numpy() == w).all()
    pt, nt, w = O.to_windows_triples_cbow(walks, ws, 40, -7, triples, 11, rng_mode=O.RNG_WALKER, window_base=99)
    gpt, gnt, gw = rw.to_windows_triples_ex(d_w, ws, -7, d_t, 11, cbow=True, base=99, rng_mode=rw.RNG_WALKER)
    assert (gpt.cpu().numpy() == pt).all() and (gnt.cpu().numpy() == nt).all() and (gw.cpu().numpy() == w).all()


@pytest.mark.gpu
@pytest.mark.parametrize("num_walks,cols,ws", [(1, 700_003, 2),      # one walk's windows overflow the tiled kernel: per-window kernel
                                               (3000, 161, 5),        # several walks per block, last block ragged
                                               (37, 2001, 3)])        # one walk per block
def test_gpu_triples_windows_shapes(num_walks, cols, ws):
    from torch_rw import rw
    rng = np.random.default_rng(cols)
    walks = rng.integers(0, 1000, (num_walks, cols))
    triples = rng.integers(0, 1000, (50, 3))
    t, p, n = O.to_windows_triples(walks, ws, 1000, -1, triples, 3)
    gt, gp, gn = rw.to_windows_triples(torch.as_tensor(walks).to(DEV), ws, 1000, -1, torch.as_tensor(triples).to(DEV), 3)
    assert (gt.cpu().numpy() == t).all() and (gp.cpu().numpy() == p).all() and (gn.cpu().numpy() == n).all()


# ---- rw.walk_windows: the walk fused with to_windows (windows_cpu.cpp:5-77 after rw_cpu.cpp:58-107) ----

def _small_graph(n, m, seed, device="cpu"):
    from torch_random_walk_amd import graphs
    return graphs.rmat(n, m, seed=seed, device=device)


def test_walk_windows_host_composition_equals_two_calls():
    """CPU tensors (and node2vec walks anywhere) run as the two calls: same tensors, optional walks."""
    from torch_rw import rw
    rp, ci = _small_graph(300, 2000, 4)
    tg = torch.arange(300).repeat(2)[:431]
    for p, q in [(1.0, 1.0), (0.5, 2.0)]:
        got = rw.walk_windows(rp, ci, tg, p, q, 20, 7, 5, 300, return_walks=True)
        walks = rw.walk(rp, ci, tg, p, q, 20, 7)
        ref = rw.to_windows(walks, 5, 300, 7)
        assert all(torch.equal(a, b) for a, b in zip(got[:3], ref)) and torch.equal(got[3], walks)
    t, pw, nw = rw.walk_windows(rp, ci, tg, 1.0, 1.0, 20, 7, 5, 300, window_seed=99)
    ref = rw.to_windows(rw.walk(rp, ci, tg, 1.0, 1.0, 20, 7), 5, 300, 99)
    assert torch.equal(nw, ref[2]) and torch.equal(pw, ref[1])
    with pytest.raises(RuntimeError):
        rw.walk_windows(rp, ci, tg, 1.0, 1.0, 20, 7, 22, 300)


def _walk_windows_follows_rng_mode(device):
    """walk_windows resolves the walk's stream as rw.walk does -- under set_rng_mode(RNG_WALKER) a uniform
    walk is the per-walker Philox stream in BOTH forms, and differs from the libc stream."""
    from torch_rw import rw
    rp, ci = _small_graph(300, 2000, 4, device)
    tg = torch.arange(300, device=device).repeat(2)[:431]
    glibc = rw.walk_windows(rp, ci, tg, 1.0, 1.0, 20, 7, 5, 300, return_walks=True)
    old = rw.get_rng_mode()
    rw.set_rng_mode(rw.RNG_WALKER)
    try:
        got = rw.walk_windows(rp, ci, tg, 1.0, 1.0, 20, 7, 5, 300, return_walks=True)
        walks = rw.walk(rp, ci, tg, 1.0, 1.0, 20, 7)
        ref = rw.to_windows(walks, 5, 300, 7)
    finally:
        rw.set_rng_mode(old)
    assert torch.equal(got[3], walks) and all(torch.equal(a, b) for a, b in zip(got[:3], ref))
    assert not torch.equal(got[3], glibc[3]) and torch.equal(got[2], glibc[2])   # other walks, same negatives


def test_walk_windows_follows_the_process_rng_mode_host():
    _walk_windows_follows_rng_mode("cpu")


@pytest.mark.gpu
def test_gpu_walk_windows_follows_the_process_rng_mode():
    _walk_windows_follows_rng_mode(DEV)


@pytest.mark.gpu
@pytest.mark.parametrize("L,ws", [(80, 5), (80, 1), (80, 2), (80, 9), (80, 10), (7, 8), (8, 9), (6, 3), (15, 4), (16, 7),
                                  (1, 2), (23, 8), (3, 4), (9, 9)])
def test_gpu_walk_windows_equals_two_calls_and_oracle(L, ws):
    """The fused kernel (window_size <= 9) and the composed path (10) against walk + to_windows on the GPU and against the
    oracle's walk + windows, for window sizes 1..10 and walk lengths on both sides of every flush boundary of the
    kernel's 16-column tile; batch sizes that leave waves partly empty; with and without the walks."""
    from torch_rw import rw
    n = 5000
    rp, ci = _small_graph(n, 40000, 6, DEV)
    tg = torch.arange(n, device=DEV).repeat(3)[:12345]
    seed = 11
    walks = rw.walk(rp, ci, tg, 1.0, 1.0, L, seed)
    ref = rw.to_windows(walks, ws, n, seed)
    for return_walks in (False, True):
        got = rw.walk_windows(rp, ci, tg, 1.0, 1.0, L, seed, ws, n, return_walks=return_walks, check=True)
        assert len(got) == (4 if return_walks else 3)
        for a, b, name in zip(got[:3], ref, ("target_nodes", "pos_windows", "neg_windows")):
            assert a.shape == b.shape and torch.equal(a, b), name
        if return_walks:
            assert torch.equal(got[3], walks)
    m = 700
    ow = O.walk(rp.cpu().numpy(), ci.cpu().numpy(), tg[:m].cpu().numpy(), 1.0, 1.0, L, seed, rng_mode=O.RNG_GLIBC_INDEXED)
    ot, op, on = O.to_windows(ow, ws, n, seed)
    k = ot.shape[0]
    assert (got[0][:k].cpu().numpy() == ot).all() and (got[1][:k].cpu().numpy() == op).all()
    assert (got[2][:k].cpu().numpy() == on).all()


@pytest.mark.gpu
def test_gpu_walk_windows_shards_and_faults():
    """walker_base numbers walkers AND windows: two shards equal the halves of one call.  A bad start node is
    reported and its windows are filled with the bad id, as the walk's row is."""
    from torch_rw import rw
    n = 3000
    rp, ci = _small_graph(n, 20000, 2, DEV)
    tg = torch.arange(n, device=DEV)
    L, ws, seed = 40, 5, 3
    full = rw.walk_windows(rp, ci, tg, 1.0, 1.0, L, seed, ws, n, return_walks=True)
    cut, per = 1701, L + 1 - ws + 1
    a = rw.walk_windows(rp, ci, tg[:cut], 1.0, 1.0, L, seed, ws, n, return_walks=True)
    b = rw.walk_windows(rp, ci, tg[cut:], 1.0, 1.0, L, seed, ws, n, walker_base=cut, return_walks=True)
    for k in range(3):
        assert torch.equal(torch.cat([a[k], b[k]]), full[k])
    assert torch.equal(torch.cat([a[3], b[3]]), full[3]) and a[0].shape[0] == cut * per
    bad = tg.clone()
    bad[77] = n + 5
    with pytest.raises(RuntimeError, match="outside"):
        rw.walk_windows(rp, ci, bad, 1.0, 1.0, L, seed, ws, n, check=True)
    got = rw.walk_windows(rp, ci, bad, 1.0, 1.0, L, seed, ws, n, check=False)
    assert (got[0][77 * per:78 * per] == n + 5).all() and (got[1][77 * per:78 * per] == n + 5).all()
    keep = torch.ones(n * per, dtype=torch.bool, device=DEV)
    keep[77 * per:78 * per] = False
    assert torch.equal(got[0][keep], full[0][keep]) and torch.equal(got[1][keep], full[1][keep])
    assert torch.equal(got[2], full[2])


@pytest.mark.gpu
def test_gpu_walk_windows_full_size():
    """VERDICT r02 next 5's shape: 1 M walks x 81 columns, window 5 (77 M windows): fused == two calls; oracle on the
    first rows."""
    from torch_rw import rw
    from torch_random_walk_amd import graphs
    n = 1_000_000
    rp, ci = graphs.rmat(n, 10_000_000, seed=1, device=DEV)
    tg = torch.arange(n, device=DEV)
    got = rw.walk_windows(rp, ci, tg, 1.0, 1.0, 80, 10, 5, n, check=True)
    walks = rw.walk(rp, ci, tg, 1.0, 1.0, 80, 10)
    ref = rw.to_windows(walks, 5, n, 10)
    for a, b in zip(got, ref):
        assert torch.equal(a, b)
    assert got[0].shape == (77 * n,) and got[1].shape == (77 * n, 4)


@pytest.mark.gpu
def test_gpu_side_stream_lanes_capture_beside_another_thread():
    """the side stream of the window builders is a pool of lanes now -- while one thread captures a
    to_windows call into a HIP graph (its lane's stream is part of that capture until it ends), another thread's
    plain call gets ANOTHER lane: it neither fails with a capture-isolation error nor lands in the first thread's
    graph, and the graph replays to the same windows afterwards.  ADVICE r04: the captured calls really do get a
    lane (an idle one joins the capture through the fork event), counted by the library."""
    import ctypes
    import threading
    from torch_rw import rw
    from torch_random_walk_amd import _native
    lib = ctypes.CDLL(_native.HIP_LIB_PATH)
    lib.trw_side_lane_counts.argtypes = [ctypes.POINTER(ctypes.c_int64)]
    lib.trw_side_lane_counts.restype = None

    def lane_counts():
        c = (ctypes.c_int64 * 2)()
        lib.trw_side_lane_counts(c)
        return int(c[0]), int(c[1])
    n = 3000
    rp, ci = _small_graph(n, 24000, 9, DEV)
    tg = torch.arange(n, device=DEV).repeat(4)
    walks = rw.walk(rp, ci, tg, 1.0, 1.0, 40, 5)
    ref = rw.to_windows(walks, 5, n, 5)                 # (also makes the first lane, outside any capture)
    other_walks = rw.walk(rp, ci, tg[:5000], 1.0, 1.0, 30, 6)
    other_ref = rw.to_windows(other_walks, 4, n, 6)
    torch.cuda.synchronize()
    errors, got_other = [], []

    def plain_call():
        try:
            s2 = torch.cuda.Stream()
            with torch.cuda.stream(s2):
                for _ in range(3):
                    got_other.append(rw.to_windows(other_walks, 4, n, 6))
            s2.synchronize()
        except Exception as e:      # noqa: BLE001
            errors.append(repr(e))

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        rw.to_windows(walks, 5, n, 5)                   # warm the allocator on the capture stream
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    before = lane_counts()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        cap = rw.to_windows(walks, 5, n, 5)
        t = threading.Thread(target=plain_call)         # runs while the capture is still open
        t.start()
        t.join()
        cap2 = rw.to_windows(walks, 5, n, 5)            # a second call of the SAME capture reuses the capture's lane
    after = lane_counts()
    assert not errors, errors
    # both captured calls forked onto a lane (the graph holds the negatives BESIDE the gather), and so did the
    # other thread's three plain calls
    assert after[1] - before[1] == 2, (before, after)
    assert after[0] - before[0] == 5, (before, after)
    for got in got_other:
        assert all(torch.equal(a, b) for a, b in zip(got, other_ref))
    for _ in range(3):
        for x in cap + cap2:
            x.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(cap, ref)) and all(torch.equal(a, b) for a, b in zip(cap2, ref))
    again = rw.to_windows(walks, 5, n, 5)               # and the lanes are usable after the capture has ended
    assert all(torch.equal(a, b) for a, b in zip(again, ref))
