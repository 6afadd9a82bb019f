"""The host leg of the dispatcher: CPU tensors through `torch_rw.rw.walk`, as the reference's README and
tests call it (tests/test_rw.py:30-55, :98-122 -> rw_init.cpp:22-23 -> walk_cpu).  The leg is product code
(torch_random_walk_amd/csrc/trw_cpu.cpp); the oracle here is only the checker."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests.helpers import hub_graph, random_graph


@pytest.fixture(scope="module")
def rw():
    from torch_rw import rw as _rw
    return _rw


def _t(*arrays):
    return [torch.as_tensor(np.asarray(a), dtype=torch.int64) for a in arrays]


def _cpu(rw, rp, ci, tg, p, q, L, seed, **kw):
    out = rw.walk_ex(*_t(rp, ci, tg), p, q, L, seed, **kw)
    assert out.device.type == "cpu" and out.dtype == torch.int64
    return out.numpy()


def test_reference_uniform_cpu_golden_through_the_reference_call(rw, golden):
    """reference tests/test_rw.py:30-55, verbatim call pattern: networkx graph -> utils.to_csr -> rw.walk on
    CPU tensors."""
    import networkx as nx
    from torch_rw import utils
    g = nx.Graph()
    for a, b in golden["graph"]["edges"]:
        g.add_edge(a, b)
    row_ptr, col_idx = utils.to_csr(g)
    nodes = utils.nodes_tensor(g)
    walks = rw.walk(row_ptr=row_ptr, col_idx=col_idx, target_nodes=nodes, p=1.0, q=1.0, walk_length=6, seed=10)
    assert walks.device.type == "cpu" and walks.dtype == torch.int64 and tuple(walks.shape) == (5, 7)
    assert walks.tolist() == golden["uniform_cpu"]["walks"]              # tests/test_rw.py:49-53


def test_reference_biased_cpu_golden_through_the_reference_call(rw, golden):
    g = golden["graph"]
    rp, ci, tg = _t(g["row_ptr"], g["col_idx"], g["nodes"])
    walks = rw.walk(row_ptr=rp, col_idx=ci, target_nodes=tg, p=0.7, q=0.5, walk_length=6, seed=10)
    assert walks.tolist() == golden["biased_cpu"]["walks"]               # tests/test_rw.py:116-120


@pytest.mark.parametrize("seed", [0, 1, 10, 123456, -1, -(2**31), 2**31 - 1])
def test_uniform_equals_oracle_seeds(rw, seed):
    rp, ci = hub_graph(400, 3)
    tg = np.random.default_rng(5).integers(0, 400, 777)
    assert (_cpu(rw, rp, ci, tg, 1.0, 1.0, 20, seed) == O.walk(rp, ci, tg, 1.0, 1.0, 20, seed)).all()


@pytest.mark.parametrize("threads", [1, 2, 3, 8])
def test_uniform_is_thread_count_invariant(rw, threads):
    """csrc/cpu is only defined on one thread; the host leg splits uniform walks by jump-ahead,
    so any thread count gives the single-thread rows."""
    rp, ci = random_graph(500, 5, 21)
    tg = np.random.default_rng(1).integers(0, 500, 5000)
    ref = O.walk(rp, ci, tg, 1.0, 1.0, 33, 10)
    before = torch.get_num_threads()
    try:
        torch.set_num_threads(threads)
        assert (_cpu(rw, rp, ci, tg, 1.0, 1.0, 33, 10) == ref).all()
    finally:
        torch.set_num_threads(before)


@pytest.mark.parametrize("p,q", [(0.25, 4.0), (0.7, 0.5), (2.0, 1.0), (1.0, 1.0 + 1e-12), (4.0, 0.25)])
def test_biased_equals_the_serial_glibc_chain(rw, p, q):
    """Beyond the reference's parallel_for grain too (N = 1000 > 32768/21): on the host AUTO is the serial
    chain for any N -- what csrc/cpu computes on one thread."""
    rp, ci = random_graph(200, 4, 51)
    tg = np.random.default_rng(6).integers(0, 200, 1000)
    assert (_cpu(rw, rp, ci, tg, p, q, 20, 10) == O.walk(rp, ci, tg, p, q, 20, 10, rng_mode=O.RNG_GLIBC_SEQ)).all()


def test_biased_hub_graph_and_directed_asymmetry(rw):
    rp, ci = hub_graph(600, 12)
    tg = np.random.default_rng(9).integers(0, 600, 300)
    assert (_cpu(rw, rp, ci, tg, 0.25, 4.0, 80, 3) == O.walk(rp, ci, tg, 0.25, 4.0, 80, 3, rng_mode=O.RNG_GLIBC_SEQ)).all()
    rp = np.array([0, 2, 4, 5], dtype=np.int64)      # is_neighbor scans t's row for x: direction matters
    ci = np.array([1, 2, 0, 2, 0], dtype=np.int64)
    tg = np.zeros(256, dtype=np.int64)
    assert (_cpu(rw, rp, ci, tg, 0.5, 2.0, 12, 3) == O.walk(rp, ci, tg, 0.5, 2.0, 12, 3, rng_mode=O.RNG_GLIBC_SEQ)).all()


@pytest.mark.parametrize("p,q", [(1.0, 1.0), (0.25, 4.0)])
def test_walker_stream_on_the_host_equals_oracle(rw, p, q):
    rp, ci = hub_graph(500, 4)
    tg = np.random.default_rng(2).integers(0, 500, 3000)
    ref = O.walk(rp, ci, tg, p, q, 40, 77, rng_mode=O.RNG_WALKER, walker_base=2**32 + 9)
    assert (_cpu(rw, rp, ci, tg, p, q, 40, 77, walker_base=2**32 + 9, rng_mode=rw.RNG_WALKER) == ref).all()


def test_uniform_walker_base_and_out(rw):
    rp, ci = random_graph(300, 4, 9)
    tg = np.random.default_rng(1).integers(0, 300, 900)
    ref = O.walk(rp, ci, tg, 1.0, 1.0, 80, 10)
    full = torch.full((900, 81), -1, dtype=torch.int64)
    for lo, hi in ((0, 1), (1, 400), (400, 900)):
        got = rw.walk_ex(*_t(rp, ci, tg[lo:hi]), 1.0, 1.0, 80, 10, walker_base=lo, out=full[lo:hi])
        assert got.data_ptr() == full[lo:hi].data_ptr()
    assert (full.numpy() == ref).all()


def test_edge_cases(rw):
    rp, ci = random_graph(50, 3, 71)
    t_rp, t_ci = _t(rp, ci)
    for p, q in ((1.0, 1.0), (0.5, 2.0)):
        assert tuple(rw.walk(t_rp, t_ci, torch.empty(0, dtype=torch.int64), p, q, 6, 1).shape) == (0, 7)
    tg = np.full(64, 7)                               # duplicates are independent walkers
    assert (_cpu(rw, rp, ci, tg, 1.0, 1.0, 12, 5) == O.walk(rp, ci, tg, 1.0, 1.0, 12, 5)).all()
    strided = (torch.arange(100) % 50)[::2]           # the reference's accessors honour strides
    assert not strided.is_contiguous()
    assert (rw.walk(t_rp, t_ci, strided, 1.0, 1.0, 9, 3).numpy() == O.walk(rp, ci, strided.numpy(), 1.0, 1.0, 9, 3)).all()
    assert (_cpu(rw, rp, ci, [3, 4, 5], 0.5, 2.0, 1, 1) == O.walk(rp, ci, [3, 4, 5], 0.5, 2.0, 1, 1,
                                                                  rng_mode=O.RNG_GLIBC_SEQ)).all()


def test_errors_are_python_exceptions(rw):
    rp, ci = random_graph(50, 3, 81)
    t_rp, t_ci, t_tg = _t(rp, ci, np.arange(50))
    with pytest.raises(RuntimeError, match="expected scalar type Long"):
        rw.walk(t_rp.to(torch.int32), t_ci, t_tg, 1.0, 1.0, 5, 1)
    with pytest.raises(RuntimeError, match="walk_length"):
        rw.walk(t_rp, t_ci, t_tg, 1.0, 1.0, 0, 1)
    rp0 = np.array([0, 1, 1, 2], dtype=np.int64)      # the reference dies with SIGFPE here (rw_cpu.cpp:20)
    ci0 = np.array([1, 0], dtype=np.int64)
    for p, q in ((1.0, 1.0), (0.5, 2.0)):
        with pytest.raises(RuntimeError, match="no out-edges"):
            rw.walk(*_t(rp0, ci0, [0, 2]), p, q, 4, 1)
    with pytest.raises(RuntimeError, match="outside"):
        rw.walk(t_rp, t_ci, torch.tensor([0, 50]), 1.0, 1.0, 4, 1)
    with pytest.raises(RuntimeError, match="outside"):
        rw.walk(t_rp, t_ci, torch.tensor([-1]), 0.5, 2.0, 4, 1)
    with pytest.raises(RuntimeError, match="monotone"):
        rw.walk(torch.tensor([0, 3, 2, 5]), torch.arange(5) % 3, torch.tensor([1]), 1.0, 1.0, 4, 1)


def test_c_abi_direct_call():
    from torch_random_walk_amd import _native
    L = _native.cabi()
    rp, ci = random_graph(300, 4, 101)
    tg = np.ascontiguousarray(np.random.default_rng(8).integers(0, 300, 500), dtype=np.int64)
    for p, q, mode, omode in ((1.0, 1.0, 1, O.RNG_GLIBC_SEQ), (0.25, 4.0, 0, O.RNG_GLIBC_SEQ), (0.25, 4.0, 2, O.RNG_WALKER)):
        out = np.empty((500, 41), dtype=np.int64)
        rc = L.trw_walk_csr_cpu(rp.ctypes.data, 300, ci.ctypes.data, len(ci), tg.ctypes.data, 500, 0, p, q, 40, 10,
                                mode, out.ctypes.data, 4)
        assert rc == 0, L.trw_last_error()
        assert (out == O.walk(rp, ci, tg, p, q, 40, 10, rng_mode=omode)).all()
    out = np.empty((2, 5), dtype=np.int64)
    rp0, ci0 = np.array([0, 1, 1], dtype=np.int64), np.array([1], dtype=np.int64)
    tg0 = np.array([0, 1], dtype=np.int64)
    rc = L.trw_walk_csr_cpu(rp0.ctypes.data, 2, ci0.ctypes.data, 1, tg0.ctypes.data, 2, 0, 1.0, 1.0, 4, 1, 0,
                            out.ctypes.data, 1)
    assert rc == -6 and b"no out-edges" in L.trw_last_error()       # TRW_E_DATA
    rc = L.trw_walk_csr_cpu(rp0.ctypes.data, 2, ci0.ctypes.data, 1, tg0.ctypes.data, 2, 5, 0.5, 2.0, 4, 1, 1,
                            out.ctypes.data, 1)
    assert rc == -4                                                  # a serial chain has no shards


def test_narrow_rows_from_the_cpu_leg():
    """walk_ex(out=<int32 tensor>): narrow rows -- the uint32 bit patterns of the same node ids -- on CPU tensors too
    (walked as int64, narrowed on the way out), and dist.wire-format round trip gives the int64 rows back."""
    from torch_rw import rw
    rp, ci = random_graph(300, 5, 2)
    tg = np.random.default_rng(2).integers(0, 300, 123)
    t_rp, t_ci, t_tg = (torch.as_tensor(a) for a in (rp, ci, tg))
    for p, q, mode in ((1.0, 1.0, rw.RNG_GLIBC), (0.25, 4.0, rw.RNG_WALKER)):
        wide = rw.walk_ex(t_rp, t_ci, t_tg, p, q, 9, 3, rng_mode=mode)
        out = torch.empty((123, 10), dtype=torch.int32)
        got = rw.walk_ex(t_rp, t_ci, t_tg, p, q, 9, 3, rng_mode=mode, out=out)
        assert got.data_ptr() == out.data_ptr() and got.dtype == torch.int32
        assert torch.equal(rw.widen_rows(got), wide) and torch.equal(got, rw.narrow_rows(wide))
    with pytest.raises(RuntimeError):
        rw.walk_ex(t_rp, t_ci, t_tg, 1.0, 1.0, 9, 3, out=torch.empty((123, 10), dtype=torch.int16))
