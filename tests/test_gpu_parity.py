"""HIP path vs the CPU oracle on a real MI355X (pytest -m gpu).

Bit-exact is the bar everywhere: this is integer index work.  Small cases are
compared element for element with the oracle; BASELINE-sized cases are checked
through size-independent properties (every step is an edge, determinism,
shard invariance) plus an oracle comparison on a sample of rows.
"""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests.helpers import hub_graph, random_graph

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def rw():
    assert torch.cuda.is_available(), "the -m gpu tier needs a GPU"
    from torch_rw import rw as _rw
    return _rw


def _dev(*arrays):
    return [torch.as_tensor(np.asarray(a), dtype=torch.int64).to(DEV) for a in arrays]


def _hip(rw, rp, ci, tg, p, q, L, seed, **kw):
    d_rp, d_ci, d_tg = _dev(rp, ci, tg)
    out = rw.walk_ex(d_rp, d_ci, d_tg, p, q, L, seed, **kw)
    return out.cpu().numpy()


# ---- the reference's own goldens through the reference's own call -------------------------

def test_readme_uniform_golden(rw, golden):
    from torch_rw import utils
    import networkx as nx
    g = nx.Graph()
    for a, b in golden["graph"]["edges"]:
        g.add_edge(a, b)
    row_ptr, col_idx = utils.to_csr(g)
    nodes = utils.nodes_tensor(g)
    c = golden["uniform_cpu"]
    walks = rw.walk(row_ptr=row_ptr.to(DEV), col_idx=col_idx.to(DEV), target_nodes=nodes.to(DEV),
                    p=1.0, q=1.0, walk_length=6, seed=10)
    assert walks.dtype == torch.int64 and walks.device.type == "cuda" and tuple(walks.shape) == (5, 7)
    assert walks.cpu().tolist() == c["walks"]          # reference tests/test_rw.py:49-53 (CPU golden)


def test_readme_biased_golden(rw, golden):
    g = golden["graph"]
    c = golden["biased_cpu"]
    d_rp, d_ci, d_tg = _dev(g["row_ptr"], g["col_idx"], g["nodes"])
    walks = rw.walk(row_ptr=d_rp, col_idx=d_ci, target_nodes=d_tg, p=0.7, q=0.5, walk_length=6, seed=10)
    assert walks.cpu().tolist() == c["walks"]          # reference tests/test_rw.py:116-120 (CPU golden)


# ---- uniform walks: literal csrc/cpu output via jump-ahead ----------------------------------

@pytest.mark.parametrize("L", [1, 2, 5, 30, 31, 32, 62, 80, 100])
def test_uniform_glibc_equals_oracle_lengths(rw, L):
    rp, ci = random_graph(500, 5, 21)
    tg = np.random.default_rng(L).integers(0, 500, 1000)
    ref = O.walk(rp, ci, tg, 1.0, 1.0, L, 10)
    assert (_hip(rw, rp, ci, tg, 1.0, 1.0, L, 10) == ref).all()


@pytest.mark.parametrize("seed", [0, 1, 10, 123456, -1, -(2**31), 2**31 - 1])
def test_uniform_glibc_equals_oracle_seeds(rw, seed):
    rp, ci = hub_graph(400, 3)
    tg = np.random.default_rng(5).integers(0, 400, 777)
    ref = O.walk(rp, ci, tg, 1.0, 1.0, 20, seed)
    assert (_hip(rw, rp, ci, tg, 1.0, 1.0, 20, seed) == ref).all()


@pytest.mark.parametrize("chain", [1, 2, 3, 4, 7])
def test_uniform_glibc_chain_lengths(rw, chain, monkeypatch):
    """Walkers chained per lane is launch geometry, never visible in the output."""
    monkeypatch.setenv("TRW_GLIBC_CHAIN", str(chain))
    rp, ci = random_graph(300, 4, 8)
    tg = np.random.default_rng(chain).integers(0, 300, 1001)
    ref = O.walk(rp, ci, tg, 1.0, 1.0, 17, 10)
    assert (_hip(rw, rp, ci, tg, 1.0, 1.0, 17, 10) == ref).all()


def test_uniform_glibc_walker_base(rw):
    rp, ci = random_graph(300, 4, 9)
    tg = np.random.default_rng(1).integers(0, 300, 900)
    ref = O.walk(rp, ci, tg, 1.0, 1.0, 80, 10)
    for lo, hi in ((0, 1), (1, 400), (400, 900)):
        got = _hip(rw, rp, ci, tg[lo:hi], 1.0, 1.0, 80, 10, walker_base=lo)
        assert (got == ref[lo:hi]).all()
    # far into the stream: more than 2^32 draws ahead
    base = 2**33 // 80 + 12345
    ref_far = O.walk(rp, ci, tg[:64], 1.0, 1.0, 80, 10, rng_mode=O.RNG_GLIBC_INDEXED, walker_base=base)
    assert (_hip(rw, rp, ci, tg[:64], 1.0, 1.0, 80, 10, walker_base=base) == ref_far).all()


def test_uniform_walker_stream_equals_oracle(rw):
    rp, ci = hub_graph(500, 4)
    tg = np.random.default_rng(2).integers(0, 500, 1500)
    for L in (1, 3, 4, 5, 80):
        ref = O.walk(rp, ci, tg, 1.0, 1.0, L, 77, rng_mode=O.RNG_WALKER, walker_base=1000)
        got = _hip(rw, rp, ci, tg, 1.0, 1.0, L, 77, walker_base=1000, rng_mode=rw.RNG_WALKER)
        assert (got == ref).all()


# ---- biased walks ------------------------------------------------------------------------

PQ = [(0.25, 4.0), (0.7, 0.5), (2.0, 1.0), (1.0, 1.0 + 1e-12), (4.0, 0.25)]


@pytest.mark.parametrize("p,q", PQ)
def test_biased_walker_equals_oracle(rw, p, q):
    rp, ci = random_graph(200, 4, 31)
    tg = np.random.default_rng(3).integers(0, 200, 2000)
    ref = O.walk(rp, ci, tg, p, q, 20, 10, rng_mode=O.RNG_WALKER)
    got = _hip(rw, rp, ci, tg, p, q, 20, 10, rng_mode=rw.RNG_WALKER)
    assert (got == ref).all()


def test_biased_walker_hub_graph_L80(rw):
    rp, ci = hub_graph(1024, 6)
    tg = np.random.default_rng(4).integers(0, 1024, 512)
    ref = O.walk(rp, ci, tg, 0.25, 4.0, 80, 10, rng_mode=O.RNG_WALKER, walker_base=2**32 + 5)
    got = _hip(rw, rp, ci, tg, 0.25, 4.0, 80, 10, rng_mode=rw.RNG_WALKER, walker_base=2**32 + 5)
    assert (got == ref).all()


@pytest.mark.parametrize("hash_path", ["0", "1"])
def test_biased_walker_unsorted_rows_take_the_scan_path(rw, hash_path, monkeypatch):
    monkeypatch.setenv("TRW_HASH", hash_path)
    rp, ci = random_graph(300, 5, 41, sort_rows=False)
    assert any((np.diff(ci[rp[v]:rp[v + 1]]) < 0).any() for v in range(300))
    tg = np.arange(300)
    ref = O.walk(rp, ci, tg, 0.5, 2.0, 25, 3, rng_mode=O.RNG_WALKER)
    got = _hip(rw, rp, ci, tg, 0.5, 2.0, 25, 3, rng_mode=rw.RNG_WALKER)
    assert (got == ref).all()


def test_biased_directed_asymmetry(rw):
    rp = np.array([0, 2, 4, 5], dtype=np.int64)
    ci = np.array([1, 2, 0, 2, 0], dtype=np.int64)
    tg = np.zeros(256, dtype=np.int64)
    ref = O.walk(rp, ci, tg, 0.5, 2.0, 12, 3, rng_mode=O.RNG_WALKER)
    assert (_hip(rw, rp, ci, tg, 0.5, 2.0, 12, 3, rng_mode=rw.RNG_WALKER) == ref).all()


@pytest.mark.parametrize("p,q", PQ)
def test_biased_glibc_serial_replay_equals_reference_semantics(rw, p, q):
    """TRW_RNG_GLIBC on a biased walk replays the reference's single serial stream."""
    rp, ci = random_graph(200, 4, 51)
    tg = np.random.default_rng(6).integers(0, 200, 300)
    ref = O.walk(rp, ci, tg, p, q, 20, 10, rng_mode=O.RNG_GLIBC_SEQ)
    got = _hip(rw, rp, ci, tg, p, q, 20, 10, rng_mode=rw.RNG_GLIBC)
    assert (got == ref).all()


@pytest.mark.parametrize("env", [{}, {"TRW_WAVE_REPLAY": "0"}, {"TRW_HASH": "0"}, {"TRW_COMMON_MAX": "0"}])
def test_biased_glibc_serial_replay_hubs_filters_and_unsorted_rows(rw, env, monkeypatch):
    """The serial chain replayed by a whole wave (64 speculative trials per round over the prepared entries) and by
    one lane over the int64 arrays: hub rows (filter masks: candidates checked against the table), long rows,
    multi-edges, unsorted rows (one-lane kernel) -- the reference's single stream either way."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    cases = [hub_graph(700, 5), random_graph(300, 150, 3), random_graph(300, 5, 4, sort_rows=False)]
    rp2 = np.array([0, 4, 6, 8], dtype=np.int64)
    ci2 = np.array([1, 1, 1, 2, 0, 2, 0, 1], dtype=np.int64)
    cases.append((rp2, ci2))
    for rp, ci in cases:
        n = len(rp) - 1
        tg = np.random.default_rng(6).integers(0, n, 120)
        for p, q in ((0.25, 4.0), (4.0, 0.25), (0.7, 0.5)):
            ref = O.walk(rp, ci, tg, p, q, 40, 10, rng_mode=O.RNG_GLIBC_SEQ)
            for _ in range(2):   # second call: prepared graph from the cache, with the neighbourhood masks
                got = _hip(rw, rp, ci, tg, p, q, 40, 10, rng_mode=rw.RNG_GLIBC)
                assert (got == ref).all()


def test_auto_mode_rule(rw):
    """AUTO: biased walks replay the reference's stream up to its parallel_for grain
    (32768/(L+1) walkers, rw_cpu.cpp:137), and use the per-walker stream beyond."""
    rp, ci = random_graph(100, 4, 61)
    L = 80
    grain = 32768 // (L + 1)
    tg = np.random.default_rng(7).integers(0, 100, grain + 1)
    small = _hip(rw, rp, ci, tg[:grain], 0.5, 2.0, L, 10)
    assert (small == O.walk(rp, ci, tg[:grain], 0.5, 2.0, L, 10, rng_mode=O.RNG_GLIBC_SEQ)).all()
    big = _hip(rw, rp, ci, tg, 0.5, 2.0, L, 10)
    assert (big == O.walk(rp, ci, tg, 0.5, 2.0, L, 10, rng_mode=O.RNG_WALKER)).all()


# ---- boundary behaviour --------------------------------------------------------------------

def test_edge_cases(rw):
    rp, ci = random_graph(50, 3, 71)
    d_rp, d_ci = _dev(rp, ci)
    empty = rw.walk(d_rp, d_ci, torch.empty(0, dtype=torch.int64, device=DEV), 1.0, 1.0, 6, 1)
    assert tuple(empty.shape) == (0, 7)
    empty_b = rw.walk(d_rp, d_ci, torch.empty(0, dtype=torch.int64, device=DEV), 0.5, 2.0, 6, 1)
    assert tuple(empty_b.shape) == (0, 7)
    # duplicates are independent walkers
    tg = np.full(64, 7)
    assert (_hip(rw, rp, ci, tg, 1.0, 1.0, 12, 5) == O.walk(rp, ci, tg, 1.0, 1.0, 12, 5)).all()
    # strided view of the targets (the reference's accessors honour strides)
    base = torch.arange(100, dtype=torch.int64, device=DEV) % 50
    strided = base[::2]
    assert not strided.is_contiguous()
    got = rw.walk(d_rp, d_ci, strided, 1.0, 1.0, 9, 3).cpu().numpy()
    assert (got == O.walk(rp, ci, strided.cpu().numpy(), 1.0, 1.0, 9, 3)).all()
    # walk_length = 1 on the biased path
    assert (_hip(rw, rp, ci, [3, 4, 5], 0.5, 2.0, 1, 1, rng_mode=rw.RNG_WALKER)
            == O.walk(rp, ci, [3, 4, 5], 0.5, 2.0, 1, 1, rng_mode=O.RNG_WALKER)).all()


@pytest.mark.parametrize("hash_path", ["0", "1"])
def test_errors_are_python_exceptions(rw, hash_path, monkeypatch):
    monkeypatch.setenv("TRW_HASH", hash_path)
    rp, ci = random_graph(50, 3, 81)
    d_rp, d_ci, d_tg = _dev(rp, ci, np.arange(50))
    with pytest.raises(RuntimeError, match="expected scalar type Long"):
        rw.walk(d_rp.to(torch.int32), d_ci, d_tg, 1.0, 1.0, 5, 1)
    with pytest.raises(RuntimeError, match="walk_length"):
        rw.walk(d_rp, d_ci, d_tg, 1.0, 1.0, 0, 1)
    with pytest.raises(RuntimeError, match="same device|CUDA tensor"):
        rw.walk(d_rp, d_ci.cpu(), d_tg, 1.0, 1.0, 5, 1)
    # degree-0 node: the reference dies with SIGFPE (rw_cpu.cpp:20); here it is an exception
    rp0 = np.array([0, 1, 1, 2], dtype=np.int64)
    ci0 = np.array([1, 0], dtype=np.int64)
    for p, q in ((1.0, 1.0), (0.5, 2.0)):
        with pytest.raises(RuntimeError, match="no out-edges"):
            rw.walk_ex(*_dev(rp0, ci0, [0, 2]), p, q, 4, 1, check=True)      # strict: synchronise, raise now
    # start node outside the graph
    with pytest.raises(RuntimeError, match="outside"):
        rw.walk_ex(d_rp, d_ci, torch.tensor([0, 50], device=DEV), 1.0, 1.0, 4, 1, check=True)
    with pytest.raises(RuntimeError, match="outside"):
        rw.walk_ex(d_rp, d_ci, torch.tensor([-1], device=DEV), 0.5, 2.0, 4, 1, check=True)
    # the launch is asynchronous and unchecked when asked to be (the reference's behaviour)
    out = rw.walk_ex(d_rp, d_ci, torch.tensor([0, 50], device=DEV), 1.0, 1.0, 4, 1, check=False)
    torch.cuda.synchronize()
    assert out[1].tolist() == [50] * 5
    rw.check_status()                                                        # nothing pending from the above


def test_runaway_rejection_loop_is_cut_and_reported(rw, monkeypatch):
    """The reference's rejection loop (rw_cpu.cpp:166-195) has no exit but an accept: on a star with p = 1e9 a walker that
    sits on a leaf may only return to the hub, with probability 1e-9 per trial.  Every kernel family cuts the loop after
    2^20 rejections at one step, takes what the last trial drew (here: the hub, the leaf's only neighbour) and reports
    TRW_ST_TRIAL_CAP -- the walk ends, the row is a valid walk, the caller hears about it."""
    from torch_random_walk_amd import _native
    lib = _native.cabi()
    n = 40
    rp = np.concatenate([[0, n - 1], n - 1 + np.arange(1, n)]).astype(np.int64)          # node 0: the hub
    ci = np.concatenate([np.arange(1, n), np.zeros(n - 1)]).astype(np.int64)
    d_rp, d_ci = _dev(rp, ci)
    tg = torch.zeros(70, dtype=torch.int64, device=DEV)                                    # every walker starts on the hub
    stream = torch.cuda.current_stream().cuda_stream
    st = (ctypes.c_int64 * 4)()
    lib.trw_walk_status.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64)]
    for what, kind in ((14, "pool"), (6, "pool"), (2, "pool without masks"), (-1, "bisection"), (0, "bisection")):
        pb = lib.trw_prepare_csr_bytes(n, len(ci), what)
        blob = torch.empty(max(pb, 16), dtype=torch.uint8, device=DEV)
        assert lib.trw_prepare_csr_hip(d_rp.data_ptr(), n, d_ci.data_ptr(), len(ci), what, blob.data_ptr(), pb, stream) == 0
        wb = lib.trw_walk_csr_prepared_workspace_bytes(70, 3, 1e9, 1.0, 2)
        ws = torch.empty(wb, dtype=torch.uint8, device=DEV)
        out = torch.full((70, 4), -7, dtype=torch.int64, device=DEV)
        rc = lib.trw_walk_csr_prepared_hip(blob.data_ptr(), pb, what, d_rp.data_ptr(), n, d_ci.data_ptr(), len(ci), tg.data_ptr(), 70, 0,
                                           1e9, 1.0, 3, 10, 2, out.data_ptr(), ws.data_ptr(), wb, stream)
        assert rc == 0, lib.trw_last_error()
        assert lib.trw_walk_status(ws.data_ptr(), stream, st) == 0
        assert st[0] == 8, (kind, what, st[0])                                            # TRW_ST_TRIAL_CAP and nothing else
        rows = out.cpu().numpy()
        assert (rows[:, 0] == 0).all() and (rows[:, 2] == 0).all(), (kind, what)
        assert ((rows[:, 1] >= 1) & (rows[:, 1] < n)).all() and ((rows[:, 3] >= 1) & (rows[:, 3] < n)).all(), (kind, what)
    with pytest.raises(RuntimeError, match="rejection loop"):
        rw.walk_ex(d_rp, d_ci, tg, 1e9, 1.0, 3, 10, rng_mode=rw.RNG_WALKER, check=True)


def test_default_call_does_not_synchronise_and_reports_faults_later(rw):
    """The default surface launches and returns like the reference's GPU entry point (rw_cuda.cu:223-245):
    a device-side data fault is read back asynchronously and raised by a LATER torch_rw call, or by
    check_status(); the faulty call itself returns its (defined) rows."""
    assert rw.get_check() == rw.CHECK_DEFERRED
    rw.check_status()
    rp0 = np.array([0, 1, 1, 2], dtype=np.int64)
    ci0 = np.array([1, 0], dtype=np.int64)
    good = _dev(*random_graph(50, 3, 81), np.arange(50))
    for p, q in ((1.0, 1.0), (0.5, 2.0)):
        # proof of "no synchronisation": a long kernel queued ahead on the same stream is still running when the
        # call has returned
        bad = _dev(rp0, ci0, [0, 2])
        spin = torch.zeros(1 << 28, dtype=torch.float32, device=DEV)
        done = torch.cuda.Event()
        torch.cuda.synchronize()
        for _ in range(40):
            spin.add_(1.0)
        w = rw.walk(*bad, p, q, 4, 1)                                        # faulty, returns at once
        done.record()
        assert not done.query(), "walk() waited for the stream"
        del spin
        torch.cuda.synchronize()
        assert w[0].tolist()[:2] == [0, 1]                                   # the rows of a faulty walker are defined
        with pytest.raises(RuntimeError, match="no out-edges"):
            rw.check_status()                                                # waits for the read-back, raises once
        rw.check_status()                                                    # reported once
        rw.walk(*good, p, q, 4, 1)
        # ... or the next call raises it
        rw.walk(*bad, p, q, 4, 1)
        torch.cuda.synchronize()
        with pytest.raises(RuntimeError, match="earlier torch_rw call.*no out-edges"):
            rw.walk(*good, p, q, 4, 1)
        rw.walk(*good, p, q, 4, 1)
    rw.set_check(True)
    try:
        with pytest.raises(RuntimeError, match="outside"):
            rw.walk(good[0], good[1], torch.tensor([50], device=DEV), 1.0, 1.0, 4, 1)
    finally:
        rw.set_check("deferred")
    rw.check_status()


def test_runs_on_the_callers_stream(rw):
    rp, ci = random_graph(200, 4, 91)
    d_rp, d_ci, d_tg = _dev(rp, ci, np.arange(200))
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        out = rw.walk(d_rp, d_ci, d_tg, 1.0, 1.0, 30, 10)
    s.synchronize()
    assert (out.cpu().numpy() == O.walk(rp, ci, np.arange(200), 1.0, 1.0, 30, 10)).all()


@pytest.mark.parametrize("ext", ["0", "1"])
def test_uniform_rows_any_alignment(rw, ext, monkeypatch):
    """The uniform kernel writes the caller's rows one 64-byte line at a time, and which columns make up
    a line depends on where the row starts: every row length modulo 8, every start offset of `out`
    modulo 64 bytes, walker counts that do not fill the last wave, chained walkers -- and never a byte
    outside the rows."""
    from torch_random_walk_amd import _native
    lib = _native.cabi()
    monkeypatch.setenv("TRW_EXT", ext)
    rp, ci = random_graph(200, 4, 33)
    d_rp, d_ci = _dev(rp, ci)
    stream = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(4)
    for L, N, chain, shift in ((1, 1, 1, 0), (2, 65, 1, 1), (6, 130, 2, 2), (7, 64, 1, 3), (8, 300, 3, 4),
                               (9, 257, 1, 5), (14, 191, 4, 6), (15, 500, 1, 7), (16, 63, 5, 1), (23, 1000, 2, 0),
                               (33, 129, 7, 3), (80, 777, 1, 5), (81, 640, 4, 2), (127, 100, 1, 7)):
        monkeypatch.setenv("TRW_GLIBC_CHAIN", str(chain))
        tg = rng.integers(0, 200, N)
        d_tg = torch.as_tensor(tg, device=DEV)
        guard = 16
        buf = torch.full((N * (L + 1) + 2 * guard + 8,), -7, dtype=torch.int64, device=DEV)
        out = buf[guard + shift: guard + shift + N * (L + 1)]
        nbytes = lib.trw_walk_csr_workspace_bytes(200, len(ci), N, L, 1.0, 1.0, 1)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
        rc = lib.trw_walk_csr_hip(d_rp.data_ptr(), 200, d_ci.data_ptr(), len(ci), d_tg.data_ptr(), N, 0, 1.0, 1.0,
                                  L, 10, 1, out.data_ptr(), ws.data_ptr(), nbytes, stream)
        assert rc == 0, lib.trw_last_error()
        ref = O.walk(rp, ci, tg, 1.0, 1.0, L, 10)
        assert (out.view(N, L + 1).cpu().numpy() == ref).all(), (L, N, chain, shift)
        assert bool((buf[:guard + shift] == -7).all()) and bool((buf[guard + shift + N * (L + 1):] == -7).all())


@pytest.mark.parametrize("width", [8, 4])
def test_pool_rows_any_alignment(rw, width):
    """The walker pool stores its columns in aligned groups of four (32 bytes of int64 columns, 16 of uint32 ones): a row's
    head and tail are ragged, and which columns share a group depends on where the row starts.  Every row length modulo 4
    from 1 up, every start offset of `out` modulo a group, batches that do and do not fill the pools, a start node
    out of range (its row is filled), against the oracle -- and never a byte outside the rows."""
    from torch_random_walk_amd import _native
    lib = _native.cabi()
    lib.trw_walk_csr_prepared_u32_hip.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64,
                                                  ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                                  ctypes.c_double, ctypes.c_double, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                                  ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    n = 300
    rp, ci = hub_graph(n, 3)
    d_rp, d_ci = _dev(rp, ci)
    nnz = len(ci)
    stream = torch.cuda.current_stream().cuda_stream
    pb = lib.trw_prepare_csr_bytes(n, nnz, 14)
    blob = torch.empty(pb, dtype=torch.uint8, device=DEV)
    assert lib.trw_prepare_csr_hip(d_rp.data_ptr(), n, d_ci.data_ptr(), nnz, 14, blob.data_ptr(), pb, stream) == 0
    rng = np.random.default_rng(9)
    dt = torch.int64 if width == 8 else torch.int32
    for L, N, shift in ((1, 70, 0), (1, 3, 1), (2, 65, 1), (2, 700, 2), (3, 130, 2), (3, 64, 3), (4, 300, 3), (5, 257, 0),
                        (6, 191, 1), (7, 500, 2), (8, 63, 3), (9, 1000, 0), (10, 129, 5), (11, 40000, 6), (12, 777, 7),
                        (80, 640, 1), (81, 100, 3)):
        tg = rng.integers(0, n, N)
        bad = N > 100
        if bad:
            tg[[7, N - 1]] = [n, -1]
        d_tg = torch.as_tensor(tg, device=DEV)
        guard = 16
        buf = torch.full((N * (L + 1) + 2 * guard + 8,), -7, dtype=dt, device=DEV)
        out = buf[guard + shift: guard + shift + N * (L + 1)]
        wb = lib.trw_walk_csr_prepared_workspace_bytes(N, L, 0.25, 4.0, 2)
        ws = torch.empty(wb, dtype=torch.uint8, device=DEV)
        entry = lib.trw_walk_csr_prepared_hip if width == 8 else lib.trw_walk_csr_prepared_u32_hip
        rc = entry(blob.data_ptr(), pb, 14, d_rp.data_ptr(), n, d_ci.data_ptr(), nnz, d_tg.data_ptr(), N, 5, 0.25, 4.0, L, 10, 2,
                   out.data_ptr(), ws.data_ptr(), wb, stream)
        assert rc == 0, lib.trw_last_error()
        torch.cuda.synchronize()
        fam = np.zeros(1, dtype=np.int64)
        ok = np.ones(N, dtype=bool)
        if bad:
            ok[[7, N - 1]] = False
        ref = O.walk(rp, ci, np.where(ok, tg, 0), 0.25, 4.0, L, 10, rng_mode=O.RNG_WALKER, walker_base=5)
        got = out.view(N, L + 1).cpu().numpy()
        if width == 4:
            got = got.view(np.uint32).astype(np.int64)
        assert (got[ok] == ref[ok]).all(), (L, N, shift)
        if bad:  # a start node out of range fills its row
            want = tg[~ok] if width == 8 else tg[~ok].astype(np.uint32).astype(np.int64)
            assert (got[~ok] == want[:, None]).all(), (L, N, shift)
        assert bool((buf[:guard + shift] == -7).all()) and bool((buf[guard + shift + N * (L + 1):] == -7).all()), (L, N, shift)


def test_uniform_rows_bad_targets_fill_their_rows(rw):
    rp, ci = random_graph(100, 3, 2)
    tg = np.arange(100).repeat(3)
    tg[[5, 64, 299]] = [100, -1, 2**40]
    d_rp, d_ci, d_tg = _dev(rp, ci, tg)
    w = rw.walk_ex(d_rp, d_ci, d_tg, 1.0, 1.0, 20, 3, check=False).cpu().numpy()
    ok = np.ones(300, dtype=bool)
    ok[[5, 64, 299]] = False
    for r in (5, 64, 299):
        assert (w[r] == tg[r]).all()
    good = tg.copy()
    good[~ok] = 0
    ref = O.walk(rp, ci, good, 1.0, 1.0, 20, 3)
    assert (w[ok] == ref[ok]).all()


# ---- through the bare C ABI ----------------------------------------------------------------

def test_c_abi_direct_call(rw):
    from torch_random_walk_amd import _native
    L = _native.cabi()
    rp, ci = random_graph(300, 4, 101)
    tg = np.random.default_rng(8).integers(0, 300, 500)
    d_rp, d_ci, d_tg = _dev(rp, ci, tg)
    for p, q, mode, omode in ((1.0, 1.0, 1, O.RNG_GLIBC_SEQ), (0.25, 4.0, 2, O.RNG_WALKER)):
        out = torch.empty((500, 41), dtype=torch.int64, device=DEV)
        nbytes = L.trw_walk_csr_workspace_bytes(300, len(ci), 500, 40, p, q, mode)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
        stream = torch.cuda.current_stream().cuda_stream
        rc = L.trw_walk_csr_hip(d_rp.data_ptr(), 300, d_ci.data_ptr(), len(ci), d_tg.data_ptr(), 500, 0, p, q,
                                40, 10, mode, out.data_ptr(), ws.data_ptr(), nbytes, stream)
        assert rc == 0, L.trw_last_error()
        st = (ctypes.c_int64 * 4)()
        assert L.trw_walk_status(ws.data_ptr(), stream, st) == 0
        assert st[0] == 0 and st[2] == mode and st[3] in (1, 2, 3, 4, 5)   # uint32-shadow kernels (2: + edge table, 3: fat entries)
        assert (out.cpu().numpy() == O.walk(rp, ci, tg, p, q, 40, 10, rng_mode=omode)).all()


VARIANTS = [{"TRW_FORCE_WIDE": "1"},                      # int64 kernels
            {"TRW_EXT": "0"},                            # uniform: compact shadow, two gathers per step
            {"TRW_EXT": "1"},                            # uniform: fat entries, one gather per step
            {"TRW_HASH": "0"},                           # bisection kernel, located-t shortcut
            {"TRW_HASH": "0", "TRW_NO_POS": "1"},        # bisection kernel, every candidate loaded
            {"TRW_HASH": "1"},                           # edge table + state machine
            {"TRW_HASH": "1", "TRW_NO_POS": "1"},
            {"TRW_HASH": "1", "TRW_RESIDENT_BLOCKS": "1"},   # persistent kernel: workgroups per CU is launch geometry
            {"TRW_HASH": "1", "TRW_RESIDENT_BLOCKS": "8", "TRW_HASH_SLACK": "12"},
            {"TRW_HASH": "1", "TRW_EXT": "1", "TRW_COMMON_MAX": "0"},      # fat entries without / with the
            {"TRW_HASH": "1", "TRW_EXT": "1", "TRW_COMMON_MAX": "4096"}]   # no-common-neighbour marks everywhere


@pytest.mark.parametrize("env", VARIANTS)
def test_kernel_variants_agree(rw, env, monkeypatch):
    """Every kernel variant of the biased walk (and the uniform walk beside it): same walks."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rp, ci = hub_graph(600, 12)
    tg = np.random.default_rng(9).integers(0, 600, 1500)
    for p, q in ((0.25, 4.0), (4.0, 0.25), (0.7, 0.5)):
        ref = O.walk(rp, ci, tg, p, q, 30, 10, rng_mode=O.RNG_WALKER)
        assert (_hip(rw, rp, ci, tg, p, q, 30, 10, rng_mode=rw.RNG_WALKER) == ref).all()
    ref = O.walk(rp, ci, tg, 1.0, 1.0, 30, 10)
    assert (_hip(rw, rp, ci, tg, 1.0, 1.0, 30, 10) == ref).all()
    # duplicates inside a row (multi-edges) keep their weight in the candidate draw
    rp2 = np.array([0, 4, 6, 8], dtype=np.int64)
    ci2 = np.array([1, 1, 1, 2, 0, 2, 0, 1], dtype=np.int64)
    tg2 = np.arange(3).repeat(100)
    ref = O.walk(rp2, ci2, tg2, 0.25, 4.0, 20, 5, rng_mode=O.RNG_WALKER)
    assert (_hip(rw, rp2, ci2, tg2, 0.25, 4.0, 20, 5, rng_mode=rw.RNG_WALKER) == ref).all()


@pytest.mark.parametrize("hash_path", ["0", "1"])
def test_work_counters(rw, hash_path, monkeypatch):
    """The counting variant of the biased kernel: same walks, trial count equal to the oracle's."""
    monkeypatch.setenv("TRW_HASH", hash_path)
    from torch_random_walk_amd import _native
    L = _native.cabi()
    rp, ci = hub_graph(500, 14)
    tg = np.random.default_rng(10).integers(0, 500, 3000)
    ref, cnt = O.walk(rp, ci, tg, 0.25, 4.0, 40, 10, rng_mode=O.RNG_WALKER, return_counters=True)
    d_rp, d_ci, d_tg = _dev(rp, ci, tg)
    out = torch.empty((3000, 41), dtype=torch.int64, device=DEV)
    nbytes = L.trw_walk_csr_workspace_bytes(500, len(ci), 3000, 40, 0.25, 4.0, 2)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    stream = torch.cuda.current_stream().cuda_stream
    L.trw_profile_enable(2)
    try:
        rc = L.trw_walk_csr_hip(d_rp.data_ptr(), 500, d_ci.data_ptr(), len(ci), d_tg.data_ptr(), 3000, 0, 0.25, 4.0,
                                40, 10, 2, out.data_ptr(), ws.data_ptr(), nbytes, stream)
        assert rc == 0, L.trw_last_error()
        c = (ctypes.c_uint64 * 4)()
        assert L.trw_walk_counters(ws.data_ptr(), stream, c) == 0
    finally:
        L.trw_profile_enable(0)
    assert (out.cpu().numpy() == ref).all()
    assert c[0] == cnt["trials_total"]
    assert c[1] <= c[0] and c[2] <= cnt["membership_tests"]      # most trials never touch memory


def test_library_selftest_of_the_draw_reduction():
    """`draw % degree` on the device (rw_cpu.cpp:20's `rand() % n`): the walker pools reduce with an exact reciprocal
    and ONE correction; the library checks that form and the two-correction one against the hardware's % for every
    divisor below 2^31."""
    from torch_random_walk_amd import _native
    L = _native.cabi()
    scratch = torch.zeros(2, dtype=torch.int64, device=DEV)
    failures = ctypes.c_uint64(12345)
    assert L.trw_selftest_hip(scratch.data_ptr(), torch.cuda.current_stream().cuda_stream, ctypes.byref(failures)) == 0, \
        L.trw_last_error()
    assert failures.value == 0


# ---- BASELINE-sized runs: properties + sampled oracle rows --------------------------------

def _check_big(rw, rp, ci, tg, p, q, L, seed, omode, sample=2000):
    from torch_random_walk_amd import graphs
    w1 = rw.walk(rp, ci, tg, p, q, L, seed)
    w2 = rw.walk(rp, ci, tg, p, q, L, seed)
    assert torch.equal(w1, w2)                                   # deterministic
    assert torch.equal(w1[:, 0], tg)                             # column 0 = start node
    assert int(w1.min()) >= 0 and int(w1.max()) < rp.numel() - 1
    assert graphs.walks_follow_edges(rp, ci, w1)                 # every step is an edge
    n = tg.numel()
    half = n // 2 + 3                                            # shard invariance
    mode = rw.RNG_GLIBC if (p == 1.0 and q == 1.0) else rw.RNG_WALKER
    a = rw.walk_ex(rp, ci, tg[:half], p, q, L, seed, walker_base=0, rng_mode=mode)
    b = rw.walk_ex(rp, ci, tg[half:], p, q, L, seed, walker_base=half, rng_mode=mode)
    assert torch.equal(torch.cat([a, b]), w1)
    # sampled rows against the oracle (global walker index = row index)
    rp_h, ci_h = rp.cpu().numpy(), ci.cpu().numpy()
    rows = np.unique(np.concatenate([[0, 1, n - 1], np.random.default_rng(0).integers(0, n, sample)]))
    tg_h = tg.cpu().numpy()
    got = w1[torch.as_tensor(rows, device=w1.device)].cpu().numpy()
    for r, g in zip(rows, got):
        ref = O.walk(rp_h, ci_h, tg_h[r:r + 1], p, q, L, seed, rng_mode=omode, walker_base=int(r))
        assert (ref[0] == g).all(), "row %d differs" % r


def test_config2_er100k_uniform_full_size(rw):
    """BASELINE config 2: Erdos-Renyi 100k nodes / 1M edges, p=q=1, L=80, 10 walks per node."""
    from torch_random_walk_amd import graphs
    rp, ci = graphs.erdos_renyi(100_000, 1_000_000, seed=1, device=DEV)
    tg = torch.arange(100_000, device=DEV).repeat(10)
    _check_big(rw, rp, ci, tg, 1.0, 1.0, 80, 10, O.RNG_GLIBC_INDEXED)


def test_config3_rmat_biased_reduced_walkers(rw):
    """BASELINE config 3's graph family at 1M nodes / 10M edges (the full 10M/100M graph runs in
    bench.py): node2vec p=0.25 q=4, L=80."""
    from torch_random_walk_amd import graphs
    rp, ci = graphs.rmat(1_000_000, 10_000_000, seed=1, device=DEV)
    tg = torch.arange(1_000_000, device=DEV)[:200_000]
    _check_big(rw, rp, ci, tg, 0.25, 4.0, 80, 10, O.RNG_WALKER, sample=300)
    _check_big(rw, rp, ci, tg, 1.0, 1.0, 80, 10, O.RNG_GLIBC_INDEXED, sample=300)


def test_config3_rmat10m_full_size(rw):
    """BASELINE config 3 at full size (the graph the north-star target is quoted on): R-MAT 10M nodes /
    100M directed edges + ring, one walker per node, L=80 -- node2vec p=0.25 q=4 on the per-walker stream
    and the uniform walk on glibc's stream.  Size-independent properties (deterministic, every step an edge,
    shard invariance) plus sampled rows against the oracle."""
    from torch_random_walk_amd import graphs
    rp, ci = graphs.rmat(10_000_000, 100_000_000, seed=1, device=DEV)
    tg = torch.arange(10_000_000, device=DEV)
    _check_big(rw, rp, ci, tg, 0.25, 4.0, 80, 10, O.RNG_WALKER, sample=48)
    _check_big(rw, rp, ci, tg, 1.0, 1.0, 80, 10, O.RNG_GLIBC_INDEXED, sample=48)


def test_more_than_2_32_output_elements(rw):
    """54 M walkers x 81 columns = 4.4e9 output elements (35 GB): every index on the way is 64-bit.
    Sampled rows (first, last, around the 2^32-element mark) against the oracle; both walks."""
    from torch_random_walk_amd import graphs
    n = 100_000
    rp, ci = graphs.erdos_renyi(n, 1_000_000, seed=3, device=DEV)
    N, L = 54_000_000, 80
    tg = torch.arange(N, device=DEV) % n
    rp_h, ci_h = rp.cpu().numpy(), ci.cpu().numpy()
    mark = (1 << 32) // (L + 1)
    rows = np.unique(np.concatenate([[0, 1, N - 2, N - 1], np.arange(mark - 3, mark + 4),
                                     np.random.default_rng(1).integers(0, N, 40)]))
    rows_d = torch.as_tensor(rows, device=DEV)
    for p, q, omode in ((1.0, 1.0, O.RNG_GLIBC_INDEXED), (0.5, 2.0, O.RNG_WALKER)):
        w = rw.walk(rp, ci, tg, p, q, L, 10)
        assert w.shape == (N, L + 1) and w.numel() > (1 << 32)
        got = w[rows_d].cpu().numpy()
        assert int(w[:, 0].ne(tg).sum()) == 0
        lo, hi = int(w.min()), int(w.max())
        del w
        torch.cuda.empty_cache()
        assert 0 <= lo and hi < n
        for r, g in zip(rows, got):
            ref = O.walk(rp_h, ci_h, np.array([r % n]), p, q, L, 10, rng_mode=omode, walker_base=int(r))
            assert (ref[0] == g).all(), "row %d differs" % r


def test_walk_is_graph_capturable(rw):
    """A call is a handful of kernels on the caller's stream, nothing else (no memset nodes: on ROCm 7.2 a
    captured hipMemsetAsync runs on the first launch of a graph only): it can be captured into a HIP graph
    and replayed any number of times -- status block, walker counter and edge table are reset by kernels
    of the call itself."""
    rp, ci = hub_graph(2000, 6)
    tg = np.random.default_rng(2).integers(0, 2000, 20_000)
    d_rp, d_ci, d_tg = _dev(rp, ci, tg)
    ref_b = O.walk(rp, ci, tg, 0.25, 4.0, 20, 10, rng_mode=O.RNG_WALKER)
    ref_u = O.walk(rp, ci, tg, 1.0, 1.0, 20, 10)
    rw.walk_ex(d_rp, d_ci, d_tg[:8], 1.0, 1.0, 20, 10)           # first call on the device uploads the jump table

    def both():
        return (rw.walk_ex(d_rp, d_ci, d_tg, 0.25, 4.0, 20, 10, rng_mode=rw.RNG_WALKER, check=False),
                rw.walk_ex(d_rp, d_ci, d_tg, 1.0, 1.0, 20, 10, check=False))

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):                                      # warm the allocator on a side stream
            both()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        wb, wu = both()
    for _ in range(3):
        wb.zero_()
        wu.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert (wb.cpu().numpy() == ref_b).all() and (wu.cpu().numpy() == ref_u).all()


def test_concurrent_calls_from_two_threads(rw):
    """The library keeps no per-call state outside the caller's workspace: two host threads walking on their
    own streams at the same time (biased + uniform, table + fat paths) get what they would get alone."""
    import threading
    rp, ci = hub_graph(3000, 7)
    d_rp, d_ci = _dev(rp, ci)
    jobs = []
    for k, (p, q, N) in enumerate(((0.25, 4.0, 30_000), (1.0, 1.0, 50_000), (4.0, 0.25, 20_000), (1.0, 1.0, 777))):
        tg = np.random.default_rng(k).integers(0, 3000, N)
        mode, omode = (rw.RNG_GLIBC, O.RNG_GLIBC_SEQ) if p == 1.0 else (rw.RNG_WALKER, O.RNG_WALKER)
        jobs.append((p, q, tg, torch.as_tensor(tg).to(DEV), mode, O.walk(rp, ci, tg, p, q, 25, 3 + k, rng_mode=omode), 3 + k))
    rw.walk_ex(d_rp, d_ci, jobs[1][3][:4], 1.0, 1.0, 25, 1)        # jump table upload happens once, up front
    errors = []

    def worker(which):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for rep in range(6):
                    p, q, tg, d_tg, mode, ref, seed = jobs[(which + rep) % len(jobs)]
                    got = rw.walk_ex(d_rp, d_ci, d_tg, p, q, 25, seed, rng_mode=mode)
                    if not (got.cpu().numpy() == ref).all():
                        errors.append("thread %d rep %d differs" % (which, rep))
        except Exception as e:      # noqa: BLE001
            errors.append("thread %d: %r" % (which, e))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_walk_into_a_callers_tensor(rw):
    """walk_ex(out=...): a shard walks straight into its slice of the gathered result."""
    rp, ci = hub_graph(500, 5)
    tg = np.random.default_rng(3).integers(0, 500, 3000)
    d_rp, d_ci, d_tg = _dev(rp, ci, tg)
    for p, q, mode, omode in ((1.0, 1.0, rw.RNG_GLIBC, O.RNG_GLIBC_INDEXED), (0.25, 4.0, rw.RNG_WALKER, O.RNG_WALKER)):
        full = torch.full((3000, 21), -1, dtype=torch.int64, device=DEV)
        a = rw.walk_ex(d_rp, d_ci, d_tg[:1000], p, q, 20, 10, walker_base=0, rng_mode=mode, out=full[:1000])
        b = rw.walk_ex(d_rp, d_ci, d_tg[1000:], p, q, 20, 10, walker_base=1000, rng_mode=mode, out=full[1000:])
        assert a.data_ptr() == full.data_ptr() and b.data_ptr() == full[1000:].data_ptr()
        assert (full.cpu().numpy() == O.walk(rp, ci, tg, p, q, 20, 10, rng_mode=omode)).all()
    with pytest.raises(RuntimeError, match="out must be"):
        rw.walk_ex(d_rp, d_ci, d_tg, 1.0, 1.0, 20, 10, out=torch.empty((3000, 20), dtype=torch.int64, device=DEV))
    with pytest.raises(RuntimeError, match="out must be"):
        rw.walk_ex(d_rp, d_ci, d_tg, 1.0, 1.0, 20, 10, out=torch.empty((3000, 42), dtype=torch.int64, device=DEV)[:, ::2])


def test_narrow_rows_from_every_kernel_family():
    """walk_ex(out=<int32 tensor>): the uint32 bit patterns of the SAME node ids the int64 call returns.  The production
    kernels write them themselves -- the reference-exact uniform walk (compact shadow and 16-byte entries), the walker
    pool (sorted rows), the state machine (unsorted rows) --; the families that write int64 only (the serial replay, the
    uniform walk on the per-walker stream, an unprepared small batch) go through a narrowing pass.  Start nodes out of
    range fill their row and report, as in the int64 call."""
    from torch_rw import rw
    from torch_random_walk_amd import _native
    rng = np.random.default_rng(12)
    rp, ci = hub_graph(4000, 3)
    urp, uci = random_graph(4000, 6, 3, sort_rows=False)
    cases = []
    for name, (a, b) in (("sorted", (rp, ci)), ("unsorted", (urp, uci))):
        d_rp, d_ci = (torch.as_tensor(x).to(DEV) for x in (a, b))
        for N, L, p, q, mode in ((50000, 30, 1.0, 1.0, rw.RNG_GLIBC),      # uniform rows kernel (fat entries: steps > nnz)
                                 (777, 13, 1.0, 1.0, rw.RNG_GLIBC),        # ... compact shadow, ragged tail
                                 (50000, 30, 0.25, 4.0, rw.RNG_WALKER),    # pool kernel / state machine on unsorted rows
                                 (300, 9, 0.25, 4.0, rw.RNG_WALKER),       # small batch, unprepared: bisection kernel
                                 (100, 20, 0.7, 0.5, rw.RNG_GLIBC),        # serial replay
                                 (5000, 8, 1.0, 1.0, rw.RNG_WALKER)):      # uniform on the per-walker stream
            cases.append((name, d_rp, d_ci, N, L, p, q, mode))
    for name, d_rp, d_ci, N, L, p, q, mode in cases:
        rw.clear_cache()
        tg = torch.as_tensor(rng.integers(0, 4000, N)).to(DEV)
        for rep in range(2):                        # (second pass: over the cached prepared graph)
            wide = rw.walk_ex(d_rp, d_ci, tg, p, q, L, 5, walker_base=rep * 1000 if mode == rw.RNG_WALKER or p == 1.0 else 0,
                              rng_mode=mode)
            out = torch.full((N, L + 1), -7, dtype=torch.int32, device=DEV)
            got = rw.walk_ex(d_rp, d_ci, tg, p, q, L, 5, walker_base=rep * 1000 if mode == rw.RNG_WALKER or p == 1.0 else 0,
                             rng_mode=mode, out=out)
            assert got.data_ptr() == out.data_ptr()
            assert torch.equal(rw.widen_rows(got), wide), (name, N, L, p, q, mode, rep)
    # the C ABI says which call shapes the walk kernel itself writes narrow rows for
    L_ = _native.cabi()
    L_.trw_walk_csr_prepared_u32_hip.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64,
                                                 ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                                 ctypes.c_double, ctypes.c_double, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                                 ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    d_rp, d_ci = (torch.as_tensor(x).to(DEV) for x in (rp, ci))
    n, nnz = 4000, len(ci)
    stream = torch.cuda.current_stream().cuda_stream
    tg = torch.arange(2000, device=DEV)
    out32 = torch.empty((2000, 11), dtype=torch.int32, device=DEV)
    for what, p, q, mode, want in ((2, 0.25, 4.0, 2, 0), (0, 1.0, 1.0, 1, 0), (1, 1.0, 1.0, 1, 0),
                                   (0, 0.25, 4.0, 2, -7), (2, 0.25, 4.0, 1, -7), (0, 1.0, 1.0, 2, -7), (-1, 0.25, 4.0, 2, -7)):
        pb = L_.trw_prepare_csr_bytes(n, nnz, what)
        blob = torch.empty(pb, dtype=torch.uint8, device=DEV)
        assert L_.trw_prepare_csr_hip(d_rp.data_ptr(), n, d_ci.data_ptr(), nnz, what, blob.data_ptr(), pb, stream) == 0
        N = 2000 if mode != 1 or p == 1.0 else 50          # (the serial replay: a small batch, walker_base 0)
        wb = L_.trw_walk_csr_prepared_workspace_bytes(N, 10, p, q, mode)
        ws = torch.empty(wb, dtype=torch.uint8, device=DEV)
        rc = L_.trw_walk_csr_prepared_u32_hip(blob.data_ptr(), pb, what, d_rp.data_ptr(), n, d_ci.data_ptr(), nnz, tg.data_ptr(), N, 0,
                                              p, q, 10, 3, mode, out32.data_ptr(), ws.data_ptr(), wb, stream)
        assert rc == want, (what, p, q, mode, rc, L_.trw_last_error())
        if rc == 0:
            wide = rw.walk_ex(d_rp, d_ci, tg[:N], p, q, 10, 3, rng_mode=mode)
            assert torch.equal(rw.widen_rows(out32[:N].contiguous()), wide)
