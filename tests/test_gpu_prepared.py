"""The prepared graph and its cache (pytest -m gpu): what `rw.walk` keeps between calls over the same tensors,
when it is invalid, and that every level of preparation walks the same rows."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests.helpers import hub_graph, random_graph

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture()
def rw():
    from torch_rw import rw as _rw
    _rw.clear_cache()
    _rw.set_cache(True, 2)
    yield _rw
    _rw.clear_cache()
    _rw.set_cache(True, 2)


def _dev(*arrays):
    return [torch.as_tensor(np.asarray(a), dtype=torch.int64).to(DEV) for a in arrays]


def test_second_walk_reuses_the_prepared_graph_and_third_has_full_marks(rw, monkeypatch):
    monkeypatch.setenv("TRW_PLAN_MARKS_AT", "1000000")   # (a call this size would bring the marks along by itself, see below)
    monkeypatch.setenv("TRW_PLAN_FULL_AT", "1000000")
    rp, ci = hub_graph(3000, 6)
    tg = np.random.default_rng(2).integers(0, 3000, 15_000)      # (30 steps per CSR entry a call)
    d_rp, d_ci, d_tg = _dev(rp, ci, tg)
    ref = O.walk(rp, ci, tg, 0.25, 4.0, 30, 10, rng_mode=O.RNG_WALKER)
    assert rw.cache_info() == []
    w1 = rw.walk(d_rp, d_ci, d_tg, 0.25, 4.0, 30, 10)
    info = rw.cache_info()
    assert len(info) == 1 and info[0][:2] == (3000, len(ci)) and info[0][2] == rw.PREP_BIASED and info[0][4] == 1
    w2 = rw.walk(d_rp, d_ci, d_tg, 0.25, 4.0, 30, 10)            # second node2vec walk (30 steps per entry behind it): + marks
    info = rw.cache_info()
    assert len(info) == 1 and info[0][2] == rw.PREP_BIASED | rw.PREP_MARKS and info[0][4] == 2
    w3 = rw.walk(d_rp, d_ci, d_tg, 0.25, 4.0, 30, 10)            # third (60 behind it): the masks of rows up to 1 024 too
    info = rw.cache_info()
    assert len(info) == 1 and info[0][2] == rw.PREP_BIASED | rw.PREP_MARKS | rw.PREP_MARKS_FULL and info[0][4] == 3
    blob_bytes = info[0][3]
    w4 = rw.walk(d_rp, d_ci, d_tg, 0.25, 4.0, 30, 10)            # served from the cache as it is
    assert rw.cache_info()[0][3] == blob_bytes and rw.cache_info()[0][4] == 4
    for w in (w1, w2, w3, w4):
        assert (w.cpu().numpy() == ref).all()
    # a uniform walk over the same graph shares the entry (its shadow); fat entries join when they pay
    u = rw.walk(d_rp, d_ci, d_tg, 1.0, 1.0, 30, 10)
    assert (u.cpu().numpy() == O.walk(rp, ci, tg, 1.0, 1.0, 30, 10)).all()
    assert len(rw.cache_info()) == 1
    # other (p, q), other batch, walker_base: the prepared graph depends on none of them
    for p, q in ((4.0, 0.25), (0.7, 0.5), (1.0, 1.0 + 1e-12)):
        got = rw.walk_ex(d_rp, d_ci, d_tg[:777], p, q, 17, 3, walker_base=5000, rng_mode=rw.RNG_WALKER)
        assert (got.cpu().numpy() == O.walk(rp, ci, tg[:777], p, q, 17, 3, rng_mode=O.RNG_WALKER, walker_base=5000)).all()
    rw.check_status()


def test_masks_of_longer_rows_join_with_the_steps_made_unless_prepared_by_hand(rw, monkeypatch):
    """The wave-per-entry marking passes are built by the first node2vec walk that finds 6 (rows up to 256) / 40 (up to
    1 024) steps per CSR entry behind it on this graph -- never behind rw.prepare()'s back; by the first walk only when
    that one call is worth the entries at all (rows up to 256 come with them) / makes 9 steps per entry itself
    (trw_prepare_plan; the tile pass made the marks that cheap)."""
    monkeypatch.setenv("TRW_PLAN_MARKS_AT", "1000000")
    monkeypatch.setenv("TRW_PLAN_FULL_AT", "1000000")
    rp, ci = hub_graph(1500, 6)
    tg = np.random.default_rng(3).integers(0, 1500, 8000)
    d_rp, d_ci, d_tg = _dev(rp, ci, tg)
    ref = O.walk(rp, ci, tg, 0.25, 4.0, 12, 4, rng_mode=O.RNG_WALKER)
    steps, nnz = len(tg) * 12, len(ci)
    n_calls = 40 * nnz // steps + 3
    seen = set()
    for k in range(n_calls):
        assert (rw.walk(d_rp, d_ci, d_tg, 0.25, 4.0, 12, 4).cpu().numpy() == ref).all()
        what = rw.cache_info()[0][2]
        behind = k * steps // nnz                                    # steps per entry made before this walk
        assert bool(what & rw.PREP_MARKS) == (k >= 1 and behind >= 6), (k, behind, what)
        assert bool(what & rw.PREP_MARKS_FULL) == (behind >= 40), (k, behind, what)
        seen.add(what)
    assert len(seen) == 3                                             # entries -> + marks -> + long-row marks
    rw.clear_cache()
    assert rw.prepare(d_rp, d_ci, marks=False) == rw.PREP_BIASED     # the caller's choice stays as it is
    for k in range(n_calls):
        rw.walk(d_rp, d_ci, d_tg, 0.25, 4.0, 12, 4)
    assert rw.cache_info()[0][2] == rw.PREP_BIASED
    rw.check_status()
    # one call that makes enough steps by itself brings the marks along (the default plan)
    monkeypatch.delenv("TRW_PLAN_MARKS_AT")
    monkeypatch.delenv("TRW_PLAN_FULL_AT")
    for walkers, want in ((nnz * 4 // 12, rw.PREP_BIASED | rw.PREP_MARKS), (nnz * 8 // 12, rw.PREP_BIASED | rw.PREP_MARKS),
                          (nnz * 10 // 12, rw.PREP_BIASED | rw.PREP_MARKS | rw.PREP_MARKS_FULL)):
        rw.clear_cache()
        t = torch.as_tensor(np.resize(tg, walkers)).to(DEV)
        got = rw.walk_ex(d_rp, d_ci, t, 0.25, 4.0, 12, 4, rng_mode=rw.RNG_WALKER)   # (2 500 walkers: AUTO would replay the serial stream)
        assert rw.cache_info()[0][2] == want, (walkers, rw.cache_info()[0][2])
        assert (got.cpu().numpy() == O.walk(rp, ci, np.resize(tg, walkers), 0.25, 4.0, 12, 4, rng_mode=O.RNG_WALKER)).all()


def test_uniform_entries_join_once_the_calls_add_up_to_the_graph(rw):
    """One uniform call smaller than the graph is not worth the 16-byte entries (trw_prepare_plan); several are: the cache
    builds them with the call at which the steps made so far reach the entry count, and the walks stay the reference's."""
    from torch_random_walk_amd import graphs
    n = 1_000_000
    d_rp, d_ci = graphs.rmat(n, 9_000_000, seed=3, device=DEV)          # 10 M entries: beyond the caches
    nnz = d_ci.numel()
    d_tg = torch.arange(0, n, 3, device=DEV)                              # 333 334 walkers x 12 = 4 M steps a call
    rp, ci, tg = d_rp.cpu().numpy(), d_ci.cpu().numpy(), d_tg.cpu().numpy()
    head = O.walk(rp, ci, tg[:64], 1.0, 1.0, 12, 5)                      # (the uniform stream is indexed by walker)
    tail = O.walk(rp, ci, tg[-64:], 1.0, 1.0, 12, 5, walker_base=len(tg) - 64)
    calls_needed = -(-nnz // (len(tg) * 12))
    assert calls_needed >= 2
    for k in range(1, calls_needed + 2):
        got = rw.walk(d_rp, d_ci, d_tg, 1.0, 1.0, 12, 5)
        has = bool(rw.cache_info()[0][2] & rw.PREP_UNIFORM)
        assert has == (k >= calls_needed), (k, calls_needed, rw.cache_info())
        assert (got[:64].cpu().numpy() == head).all() and (got[-64:].cpu().numpy() == tail).all()
    rw.check_status()


def test_in_place_write_invalidates(rw):
    rp, ci = random_graph(400, 6, 3)
    tg = np.arange(400).repeat(20)
    d_rp, d_ci, d_tg = _dev(rp, ci, tg)
    for p, q, omode in ((0.25, 4.0, O.RNG_WALKER), (1.0, 1.0, O.RNG_GLIBC_SEQ)):
        mode = rw.RNG_WALKER if omode == O.RNG_WALKER else rw.RNG_GLIBC
        for _ in range(3):
            rw.walk_ex(d_rp, d_ci, d_tg, p, q, 20, 7, rng_mode=mode)
        # rewire: every row's entries reversed (rows stay sorted? no -- so re-sort each row descending->ascending of n-1-x)
        ci2 = ci.copy()
        n = 400
        for v in range(n):
            ci2[rp[v]:rp[v + 1]] = np.sort((ci[rp[v]:rp[v + 1]] * 7 + 3) % n)
        d_ci.copy_(torch.as_tensor(ci2).to(DEV))                 # in place: same storage, new version
        got = rw.walk_ex(d_rp, d_ci, d_tg, p, q, 20, 7, rng_mode=mode).cpu().numpy()
        assert (got == O.walk(rp, ci2, tg, p, q, 20, 7, rng_mode=omode)).all(), "stale prepared graph was used"
        d_ci.copy_(torch.as_tensor(ci).to(DEV))
        got = rw.walk_ex(d_rp, d_ci, d_tg, p, q, 20, 7, rng_mode=mode).cpu().numpy()
        assert (got == O.walk(rp, ci, tg, p, q, 20, 7, rng_mode=omode)).all()


def test_entries_die_with_their_tensors_and_lru(rw):
    graphs = [random_graph(300 + 10 * k, 5, k) for k in range(4)]
    devs = [_dev(rp, ci) for rp, ci in graphs]
    tg = torch.arange(300, device=DEV).repeat(4)
    def walk(d):      # (explicit per-walker stream: under AUTO a batch this small is the serial replay, which prepares nothing)
        return rw.walk_ex(d[0], d[1], tg, 0.5, 2.0, 10, 1, rng_mode=rw.RNG_WALKER)

    for k in range(3):
        walk(devs[k])
    assert [i[0] for i in rw.cache_info()] == [320, 310]                  # capacity 2, most recent first
    rw.set_cache(True, 3)
    walk(devs[0])
    assert [i[0] for i in rw.cache_info()] == [300, 320, 310]
    del devs[2]
    walk(devs[0])                                                         # the dead graph's entry goes at the next call
    assert [i[0] for i in rw.cache_info()] == [300, 310]
    # a NEW graph in freed-and-reused memory never matches an old entry: the key holds weak storage references
    rp, ci = graphs[1]
    ptr = devs[1][1].data_ptr()
    del devs[1]
    torch.cuda.synchronize()
    ci_new = ((torch.as_tensor(ci) + 1) % 310).to(DEV)
    rp_new = torch.as_tensor(rp).to(DEV)
    ci_sorted = ci_new.cpu().numpy().copy()
    for v in range(310):
        ci_sorted[rp[v]:rp[v + 1]] = np.sort(ci_sorted[rp[v]:rp[v + 1]])
    ci_new.copy_(torch.as_tensor(ci_sorted))
    got = rw.walk_ex(rp_new, ci_new, tg, 0.5, 2.0, 10, 1, rng_mode=rw.RNG_WALKER).cpu().numpy()
    assert (got == O.walk(rp, ci_sorted, tg.cpu().numpy(), 0.5, 2.0, 10, 1, rng_mode=O.RNG_WALKER)).all()
    assert ptr  # (whether the allocator reused the address or not, the walk above had to be right)
    rw.clear_cache()
    assert rw.cache_info() == []


@pytest.mark.parametrize("deg,inline_cap", [(5, None), (60, None), (60, "0"), (150, None), (150, "4096"), (150, "0")])
def test_every_marking_path_walks_the_same_rows(rw, deg, inline_cap, monkeypatch):
    """Masks worked out by the owning thread (short rows), by the tile pass (rows of 33..4096), exact
    (rows <= 128) and as filters (longer rows: set bits ask the table), or not at all (cap 0 without marks)."""
    monkeypatch.setenv("TRW_PLAN_MARKS_AT", "1000000")          # (first walk: entries only; second: the cache adds the marks)
    monkeypatch.setenv("TRW_PLAN_FULL_AT", "1000000")
    if inline_cap is not None:
        monkeypatch.setenv("TRW_COMMON_MAX", inline_cap)
    rp, ci = random_graph(400, deg, 11)
    tg = np.random.default_rng(1).integers(0, 400, 20000)       # (enough steps per entry for the second walk to add the marks)
    d_rp, d_ci, d_tg = _dev(rp, ci, tg)
    for p, q in ((0.25, 4.0), (4.0, 0.25)):
        ref = O.walk(rp, ci, tg, p, q, 25, 9, rng_mode=O.RNG_WALKER)
        rw.clear_cache()
        cold = rw.walk_ex(d_rp, d_ci, d_tg, p, q, 25, 9, rng_mode=rw.RNG_WALKER)          # cheap marks
        warm = rw.walk_ex(d_rp, d_ci, d_tg, p, q, 25, 9, rng_mode=rw.RNG_WALKER)          # full marks (second walk)
        assert rw.cache_info()[0][2] & rw.PREP_MARKS
        assert (cold.cpu().numpy() == ref).all() and (warm.cpu().numpy() == ref).all()
    rw.clear_cache()
    assert rw.prepare(d_rp, d_ci) == rw.PREP_BIASED | rw.PREP_MARKS
    assert (rw.walk(d_rp, d_ci, d_tg, 0.25, 4.0, 25, 9).cpu().numpy() == O.walk(rp, ci, tg, 0.25, 4.0, 25, 9, rng_mode=O.RNG_WALKER)).all()
    assert rw.prepare(d_rp, d_ci, biased=False, marks=False, uniform=True) == rw.PREP_BIASED | rw.PREP_MARKS | rw.PREP_UNIFORM
    full = rw.PREP_BIASED | rw.PREP_MARKS | rw.PREP_MARKS_FULL | rw.PREP_UNIFORM
    assert rw.prepare(d_rp, d_ci, marks="full") == full
    for p, q in ((0.25, 4.0), (4.0, 0.25)):
        assert (rw.walk_ex(d_rp, d_ci, d_tg, p, q, 25, 9, rng_mode=rw.RNG_WALKER).cpu().numpy()
                == O.walk(rp, ci, tg, p, q, 25, 9, rng_mode=O.RNG_WALKER)).all()
    assert rw.cache_info()[0][2] == full


@pytest.mark.parametrize("env", [{}, {"TRW_POOL_KERNEL": "0"}, {"TRW_HASH_SLACK": "12", "TRW_COMMON_MAX": "0"}])
def test_parallel_edges_beyond_the_packed_run_and_long_probe_chains(rw, env, monkeypatch):
    """An entry packs "where t sits in prev's row" as multiplicity (4 bits) and first offset: pairs joined by more
    than 15 parallel edges go through the table instead, and the walker pool then keeps the run where two of its
    mask words were.  With an edge set at load 0.8 and no masks every in-band trial is a lookup, some of them
    with probe chains longer than the pool's 3-bit probe counter."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rng = np.random.default_rng(5)
    n = 300
    src = [rng.integers(0, n, 4000)]
    dst = [rng.integers(0, n, 4000)]
    for a, b, m_ab, m_ba in ((0, 1, 40, 40), (2, 3, 16, 16), (4, 5, 15, 15), (6, 7, 17, 1), (8, 9, 1, 300), (10, 10, 20, 0)):
        src += [np.full(m_ab, a), np.full(m_ba, b)]
        dst += [np.full(m_ab, b), np.full(m_ba, a)]
    src.append(np.arange(n))                 # no dead ends
    dst.append((np.arange(n) + 1) % n)
    src, dst = np.concatenate(src), np.concatenate(dst)
    order = np.lexsort((dst, src))
    src, dst = src[order], dst[order]
    rp = np.zeros(n + 1, dtype=np.int64)
    np.add.at(rp, src + 1, 1)
    rp = np.cumsum(rp)
    ci = dst.astype(np.int64)
    tg = np.concatenate([np.arange(12).repeat(300), rng.integers(0, n, 3000)])
    d = _dev(rp, ci, tg)
    for p, q in ((0.25, 4.0), (4.0, 0.25), (0.5, 1.0)):
        ref = O.walk(rp, ci, tg, p, q, 30, 3, rng_mode=O.RNG_WALKER)
        for _ in range(3):                   # cold, marks, served from the cache
            assert (rw.walk_ex(*d, p, q, 30, 3, rng_mode=rw.RNG_WALKER).cpu().numpy() == ref).all()
    rw.check_status()


@pytest.mark.parametrize("env", [{"TRW_MASK_KERNEL": "0"}, {"TRW_MASK_KERNEL": "0", "TRW_NO_POS": "1"},
                                 {"TRW_RESIDENT_BLOCKS": "1"}, {"TRW_HASH_SLACK": "12", "TRW_RESIDENT_BLOCKS": "8"}])
def test_mask_kernel_and_state_machine_kernel_agree(rw, env, monkeypatch):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rp, ci = hub_graph(800, 5)
    tg = np.random.default_rng(9).integers(0, 800, 4000)
    d = _dev(rp, ci, tg)
    for p, q in ((0.25, 4.0), (4.0, 0.25), (0.7, 0.5)):
        ref = O.walk(rp, ci, tg, p, q, 30, 10, rng_mode=O.RNG_WALKER)
        for _ in range(2):
            assert (rw.walk_ex(*d, p, q, 30, 10, rng_mode=rw.RNG_WALKER).cpu().numpy() == ref).all()


def test_cache_off_and_capture_bypass(rw):
    rp, ci = hub_graph(2000, 6)
    tg = np.random.default_rng(2).integers(0, 2000, 20_000)
    d_rp, d_ci, d_tg = _dev(rp, ci, tg)
    ref = O.walk(rp, ci, tg, 0.25, 4.0, 20, 10, rng_mode=O.RNG_WALKER)
    rw.set_cache(False)
    assert (rw.walk(d_rp, d_ci, d_tg, 0.25, 4.0, 20, 10).cpu().numpy() == ref).all() and rw.cache_info() == []
    rw.set_cache(True)
    rw.walk_ex(d_rp, d_ci, d_tg[:8], 1.0, 1.0, 20, 10)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            rw.walk_ex(d_rp, d_ci, d_tg, 0.25, 4.0, 20, 10, rng_mode=rw.RNG_WALKER, check=False)
    torch.cuda.current_stream().wait_stream(s)
    before = rw.cache_info()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        wb = rw.walk_ex(d_rp, d_ci, d_tg, 0.25, 4.0, 20, 10, rng_mode=rw.RNG_WALKER, check=False)
    assert rw.cache_info() == before                              # a captured call prepares for itself
    rw.clear_cache()                                              # ... so the graph survives the cache being dropped
    for _ in range(2):
        wb.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert (wb.cpu().numpy() == ref).all()


def test_prepared_graph_from_another_stream_and_thread(rw):
    import threading
    rp, ci = hub_graph(3000, 7)
    d_rp, d_ci = _dev(rp, ci)
    tg = np.random.default_rng(4).integers(0, 3000, 30_000)
    d_tg = torch.as_tensor(tg).to(DEV)
    ref = O.walk(rp, ci, tg, 0.25, 4.0, 25, 3, rng_mode=O.RNG_WALKER)
    errors = []

    def worker(k):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for _ in range(5):                                # first arrivals build / upgrade, the others reuse
                    got = rw.walk_ex(d_rp, d_ci, d_tg, 0.25, 4.0, 25, 3, rng_mode=rw.RNG_WALKER)
                    if not (got.cpu().numpy() == ref).all():
                        errors.append("thread %d differs" % k)
        except Exception as e:      # noqa: BLE001
            errors.append("thread %d: %r" % (k, e))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert len(rw.cache_info()) == 1 and rw.cache_info()[0][4] == 15


def test_prepared_c_abi_direct(rw):
    """trw_prepare_csr_hip + trw_walk_csr_prepared_hip with raw pointers, every `what`."""
    from torch_random_walk_amd import _native
    L = _native.cabi()
    rp, ci = hub_graph(700, 8)
    tg = np.random.default_rng(8).integers(0, 700, 3000)
    d_rp, d_ci, d_tg = _dev(rp, ci, tg)
    stream = torch.cuda.current_stream().cuda_stream
    n, nnz, N, Lw = 700, len(ci), 3000, 30
    for what in (0, 1, 2, 6, 7, 14, -1, -2):
        pb = L.trw_prepare_csr_bytes(n, nnz, what)
        blob = torch.empty(pb, dtype=torch.uint8, device=DEV)
        assert L.trw_prepare_csr_hip(d_rp.data_ptr(), n, d_ci.data_ptr(), nnz, what, blob.data_ptr(), pb, stream) == 0
        for p, q, mode, omode in ((1.0, 1.0, 1, O.RNG_GLIBC_SEQ), (0.25, 4.0, 2, O.RNG_WALKER), (1.0, 1.0, 2, O.RNG_WALKER)):
            out = torch.empty((N, Lw + 1), dtype=torch.int64, device=DEV)
            wb = L.trw_walk_csr_prepared_workspace_bytes(N, Lw, p, q, mode)
            ws = torch.empty(wb, dtype=torch.uint8, device=DEV)
            rc = L.trw_walk_csr_prepared_hip(blob.data_ptr(), pb, what, d_rp.data_ptr(), n, d_ci.data_ptr(), nnz,
                                             d_tg.data_ptr(), N, 0, p, q, Lw, 10, mode, out.data_ptr(), ws.data_ptr(),
                                             wb, stream)
            if what == -2 and p != 1.0:                            # no row-order probe: refused for a biased walk
                assert rc == -1
                continue
            assert rc == 0, L.trw_last_error()
            st = (ctypes.c_int64 * 4)()
            assert L.trw_walk_status(ws.data_ptr(), stream, st) == 0 and st[0] == 0
            assert (out.cpu().numpy() == O.walk(rp, ci, tg, p, q, Lw, 10, rng_mode=omode)).all(), (what, p, q, mode)
            if what in (2, 6, 7, 14) and mode == 2 and p != 1.0:
                assert st[3] in (4, 5)                             # the mask / walker-pool kernel


def _prepared_ext32(L, d_rp, d_ci, n, nnz, what):
    """The 32-byte entries of a freshly prepared graph as an [nnz, 8] uint32 array (C ABI, raw pointers)."""
    pb = L.trw_prepare_csr_bytes(n, nnz, what)
    blob = torch.empty(pb, dtype=torch.uint8, device=DEV)
    stream = torch.cuda.current_stream().cuda_stream
    assert L.trw_prepare_csr_hip(d_rp.data_ptr(), n, d_ci.data_ptr(), nnz, what, blob.data_ptr(), pb, stream) == 0
    off, size = ctypes.c_size_t(), ctypes.c_size_t()
    assert L.trw_prepare_csr_region(n, nnz, what, 5, ctypes.byref(off), ctypes.byref(size)) == 0 and size.value == nnz * 32
    torch.cuda.synchronize()
    return blob[off.value:off.value + size.value].cpu().numpy().view(np.uint32).reshape(nnz, 8)


def _skewed_graph(n, m, seed, hubs=()):
    """R-MAT-like skew (ids drawn as products of uniforms) + a few hubs of given out-degree: rows from 1 to several
    thousand entries, plenty of common neighbours among the long rows, a sprinkling of parallel edges."""
    rng = np.random.default_rng(seed)
    src = (n * rng.random(m) ** 3).astype(np.int64)
    dst = (n * rng.random(m) ** 3).astype(np.int64)
    extra_s, extra_d = [np.arange(n)], [(np.arange(n) + 1) % n]
    for h, deg in hubs:
        extra_s.append(np.full(deg, h))
        extra_d.append((n * rng.random(deg) ** 2).astype(np.int64))
    src, dst = np.concatenate([src] + extra_s), np.concatenate([dst] + extra_d)
    order = np.lexsort((dst, src))
    src, dst = src[order], dst[order]
    row_ptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(src, minlength=n), out=row_ptr[1:])
    return row_ptr, dst.astype(np.int64)


def _prepared_header(L, d_rp, d_ci, n, nnz, what):
    """The 256-byte header of a freshly prepared graph as 32 int64 slots."""
    pb = L.trw_prepare_csr_bytes(n, nnz, what)
    blob = torch.empty(pb, dtype=torch.uint8, device=DEV)
    stream = torch.cuda.current_stream().cuda_stream
    assert L.trw_prepare_csr_hip(d_rp.data_ptr(), n, d_ci.data_ptr(), nnz, what, blob.data_ptr(), pb, stream) == 0
    torch.cuda.synchronize()
    return blob[:256].cpu().numpy().view(np.int64)


@pytest.mark.parametrize("what", [6, 14])
def test_tile_pass_masks_equal_numpy(rw, what):
    """The neighbourhood masks of the entries whose shorter row is longer than 32: worked out by the tile pass (the
    longer row in LDS behind a Bloom filter, in-lists for the rows entered from shorter ones) and by numpy -- the same 32
    bytes for every sampled entry.  The graph has rows on both sides of every threshold: 8 (the owning thread / the tile
    pass), 128 (exact mask / filter), 255 (one-wave tiles / workgroup tiles), 256 / 1024 (the caps), and hubs beyond the
    4 096-entry tile."""
    from torch_random_walk_amd import _native
    L = _native.cabi()
    n = 40000
    rp, ci = _skewed_graph(n, 500_000, 5, hubs=((3, 17000), (11, 20000), (40, 5000), (41, 1100), (77, 300)))
    nnz = len(ci)
    d_rp, d_ci = _dev(rp, ci)
    tiled = _prepared_ext32(L, d_rp, d_ci, n, nnz, what)
    # numpy: entries whose shorter row has 33 .. cap entries carry the exact intersection (modulo 128 positions)
    cap = 1024 if what & 8 else 256
    deg = np.diff(rp)
    src = np.repeat(np.arange(n), deg)
    rng = np.random.default_rng(1)
    heavy = np.nonzero((np.minimum(deg[src], deg[ci]) > 32) & (np.minimum(deg[src], deg[ci]) <= cap))[0]
    assert heavy.size > 1000
    a_side = heavy[deg[src[heavy]] > deg[ci[heavy]]]
    b_side = heavy[deg[src[heavy]] <= deg[ci[heavy]]]
    assert a_side.size > 100 and b_side.size > 100
    for e in np.concatenate([rng.choice(a_side, 150), rng.choice(b_side, 150), heavy[:20], heavy[-20:]]):
        u, v = src[e], ci[e]
        row_u, row_v = ci[rp[u]:rp[u + 1]], ci[rp[v]:rp[v + 1]]
        bits = np.zeros(128, dtype=bool)
        np.logical_or.at(bits, np.nonzero(np.isin(row_v, row_u))[0] % 128, True)
        want = np.packbits(bits, bitorder="little").view(np.uint32)
        assert (tiled[e, 4:] == want).all(), (e, u, v, deg[u], deg[v])
        assert tiled[e, 0] == v and tiled[e, 1] == rp[v] and (tiled[e, 2] & 0x7FFFFFFF) == deg[v]
        assert bool(tiled[e, 2] >> 31) == (deg[v] > 128)
    # the entries of short rows (the owning thread's: rows of u of up to 8 entries -- compared in registers or asked through
    # the edge set -- and the short rows of v that ride along in u's tile) and rpos -- where u sits in v's row:
    # multiplicity << 28 | first offset, 1 = "too wide", 0 = absent
    light = np.nonzero(np.minimum(deg[src], deg[ci]) <= 32)[0]
    recip = light[np.isin(ci[light] * n + src[light], src * n + ci)]          # entries whose reverse edge exists
    assert recip.size > 200
    for e in np.concatenate([rng.choice(light, 300), rng.choice(recip, 200), rng.choice(heavy, 100)]):
        u, v = src[e], ci[e]
        row_u, row_v = ci[rp[u]:rp[u + 1]], ci[rp[v]:rp[v + 1]]
        if min(deg[u], deg[v]) <= 32:
            bits = np.zeros(128, dtype=bool)
            np.logical_or.at(bits, np.nonzero(np.isin(row_v, row_u))[0] % 128, True)
            assert (tiled[e, 4:] == np.packbits(bits, bitorder="little").view(np.uint32)).all(), (e, u, v, deg[u], deg[v])
        at = np.nonzero(row_v == u)[0]
        want_rpos = 0 if at.size == 0 else ((at.size << 28) | int(at[0]) if at.size <= 15 else 1)
        assert tiled[e, 3] == want_rpos, (e, u, v, tiled[e, 3], want_rpos)


def _numpy_entry(rp, ci, src, e):
    """What the 32-byte entry e must hold: (v, start(v), deg(v), rpos, 128-bit mask as 4 words)."""
    u, v = src[e], ci[e]
    row_u, row_v = ci[rp[u]:rp[u + 1]], ci[rp[v]:rp[v + 1]]
    bits = np.zeros(128, dtype=bool)
    np.logical_or.at(bits, np.nonzero(np.isin(row_v, row_u))[0] % 128, True)
    at = np.nonzero(row_v == u)[0]
    rpos = 0 if at.size == 0 else ((at.size << 28) | int(at[0]) if at.size <= 15 else 1)
    return v, rp[v], row_v.size, rpos, np.packbits(bits, bitorder="little").view(np.uint32)


def test_every_edge_from_a_shorter_row_to_a_longer_one_nothing_dropped(rw):
    """VERDICT r04 #6 / ADVICE r04: the in-lists of the tile pass hold the entries u -> v whose row of u is the shorter
    one.  Round 4 had room for 3/5 of the entries and silently left the rest their all-ones masks; on THIS graph -- every
    edge points from a row of 9 .. 60 entries at a row of 100 .. 700 -- every entry is such an item.  The arrays are
    sized for the worst case now: the header counts nnz items and 0 dropped, no entry whose shorter row is within the
    cap keeps an all-ones mask, sampled entries equal numpy, and the walk over it equals the oracle's."""
    from torch_random_walk_amd import _native
    L = _native.cabi()
    rng = np.random.default_rng(21)
    n_low, n_hub = 6000, 60
    n = n_low + n_hub
    rows = []
    for u in range(n_low):                                  # short rows: only hubs in them
        rows.append(np.sort(rng.choice(np.arange(n_low, n), rng.integers(9, 61), replace=True)))
    for h in range(n_hub):                                  # long rows: short-row nodes and other hubs, LONGER than any short row
        k = int(rng.integers(100, 701))
        rows.append(np.sort(np.concatenate([rng.choice(n_low, k - 5), rng.choice(np.arange(n_low, n), 5)])))
    deg = np.array([r.size for r in rows])
    rp = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(deg, out=rp[1:])
    ci = np.concatenate(rows).astype(np.int64)
    nnz = len(ci)
    src = np.repeat(np.arange(n), deg)
    low_to_hub = np.nonzero(src < n_low)[0]
    assert (deg[src[low_to_hub]] < deg[ci[low_to_hub]]).all() and low_to_hub.size > 0.85 * nnz
    d_rp, d_ci = _dev(rp, ci)
    for what in (6, 14):
        hdr = _prepared_header(L, d_rp, d_ci, n, nnz, what)
        items_expected = int(((deg[src] > 8) & (deg[src] <= (1024 if what & 8 else 256)) &
                              ((deg[src] < deg[ci]) | ((deg[src] == deg[ci]) & (src >= ci)))).sum())
        assert hdr[19] == 0, "in-list items or tasks dropped"
        assert hdr[20] == items_expected and items_expected >= low_to_hub.size
        ext = _prepared_ext32(L, d_rp, d_ci, n, nnz, what)
        cap = 1024 if what & 8 else 256
        within = np.minimum(deg[src], deg[ci]) <= cap
        # a computed mask of this graph is never all ones (no row holds 128 common neighbours with another's; a hub's
        # self-loop does, with itself)
        assert not (ext[within & (src != ci), 4:] == 0xFFFFFFFF).all(axis=1).any()
        for e in rng.choice(nnz, 1500, replace=False):
            v, start, dv, rpos, mask = _numpy_entry(rp, ci, src, e)
            assert ext[e, 0] == v and ext[e, 1] == start and (ext[e, 2] & 0x7FFFFFFF) == dv and ext[e, 3] == rpos
            if within[e]:
                assert (ext[e, 4:] == mask).all() and bool(ext[e, 2] >> 31) == (dv > 128), (what, e)
    tg = rng.integers(0, n, 4000)
    rw.clear_cache()
    assert rw.prepare(d_rp, d_ci, marks="full") & rw.PREP_MARKS_FULL
    got = rw.walk_ex(d_rp, d_ci, torch.as_tensor(tg).to(DEV), 0.25, 4.0, 20, 7, rng_mode=rw.RNG_WALKER)
    assert (got.cpu().numpy() == O.walk(rp, ci, tg, 0.25, 4.0, 20, 7, rng_mode=O.RNG_WALKER)).all()


def test_unsorted_rows_with_marks_write_nothing_outside_the_prepared_blob(rw):
    """on a graph whose rows are NOT sorted the plan kernel makes no in-list segments -- and round 4's
    scatter kernel then read its segment offsets from memory nobody had written, an 8-byte store per entry up to 32 GB past
    the item array.  Rows of 9 .. 300 entries in random order, TRW_PREP_MARKS, the blob poisoned with 0x5A beforehand and
    fenced by guard regions on both sides: the guards survive, every entry says "ask the edge set" (all ones + filter),
    and the walk equals the oracle's."""
    from torch_random_walk_amd import _native
    L = _native.cabi()
    rng = np.random.default_rng(4)
    n = 5000
    deg = rng.integers(9, 301, n)
    deg[::7] = rng.integers(1, 9, deg[::7].size)
    rp = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(deg, out=rp[1:])
    ci = rng.integers(0, n, rp[-1]).astype(np.int64)        # unsorted rows, parallel edges
    nnz = len(ci)
    d_rp, d_ci = _dev(rp, ci)
    stream = torch.cuda.current_stream().cuda_stream
    guard = 1 << 20
    for what in (6, 14):
        pb = L.trw_prepare_csr_bytes(n, nnz, what)
        pb_al = (pb + 255) // 256 * 256
        buf = torch.full((guard + pb_al + guard,), 0x5A, dtype=torch.uint8, device=DEV)
        blob = buf[guard:guard + pb_al]
        assert L.trw_prepare_csr_hip(d_rp.data_ptr(), n, d_ci.data_ptr(), nnz, what, blob.data_ptr(), pb, stream) == 0
        torch.cuda.synchronize()
        assert bool((buf[:guard] == 0x5A).all()) and bool((buf[guard + pb_al:] == 0x5A).all()), "a write outside the blob"
        hdr = blob[:256].cpu().numpy().view(np.int64)
        assert hdr[8] != hdr[9]                              # descents inside rows: not sorted
        assert hdr[16] == 0 and hdr[21] == 0 and hdr[20] == 0 and hdr[19] == 0      # no tasks, no items
        off, size = ctypes.c_size_t(), ctypes.c_size_t()
        assert L.trw_prepare_csr_region(n, nnz, what, 5, ctypes.byref(off), ctypes.byref(size)) == 0
        ext = blob[off.value:off.value + size.value].cpu().numpy().view(np.uint32).reshape(nnz, 8)
        assert (ext[:, 4:] == 0xFFFFFFFF).all() and (ext[:, 2] >> 31).all()
        tg = rng.integers(0, n, 2000)
        d_tg = torch.as_tensor(tg).to(DEV)
        out = torch.empty((2000, 13), dtype=torch.int64, device=DEV)
        wb = L.trw_walk_csr_prepared_workspace_bytes(2000, 12, 0.25, 4.0, 2)
        ws = torch.empty(wb, dtype=torch.uint8, device=DEV)
        assert L.trw_walk_csr_prepared_hip(blob.data_ptr(), pb, what, d_rp.data_ptr(), n, d_ci.data_ptr(), nnz,
                                           d_tg.data_ptr(), 2000, 0, 0.25, 4.0, 12, 3, 2, out.data_ptr(), ws.data_ptr(), wb,
                                           stream) == 0, L.trw_last_error()
        assert (out.cpu().numpy() == O.walk(rp, ci, tg, 0.25, 4.0, 12, 3, rng_mode=O.RNG_WALKER)).all()


def test_prepare_is_repeatable_entry_for_entry(rw):
    """The marking passes take their tasks and in-list places in whatever order the hardware hands them out; the 32-byte
    entries must not notice.  Five prepares of one graph -- hubs on both sides of the 4 096-entry tile, so that whole-row
    tiles, sampled ones and the one-wave kernel all run -- into buffers poisoned differently each time: identical
    entries.  (Round 5 caught a survivor-search that took a lane beyond the survivors for a hit when the LDS it read
    happened to hold 0xFFFFFFFF: one wrong mask bit in a prepare out of a dozen, found by bench.py's re-walk check.)"""
    from torch_random_walk_amd import _native
    L = _native.cabi()
    n = 60000
    rp, ci = _skewed_graph(n, 900_000, 9, hubs=((2, 30000), (5, 9000), (9, 4300), (10, 4200), (31, 4100), (50, 2000), (64, 600)))
    nnz = len(ci)
    d_rp, d_ci = _dev(rp, ci)
    stream = torch.cuda.current_stream().cuda_stream
    off, size = ctypes.c_size_t(), ctypes.c_size_t()
    assert L.trw_prepare_csr_region(n, nnz, 14, 5, ctypes.byref(off), ctypes.byref(size)) == 0
    pb = L.trw_prepare_csr_bytes(n, nnz, 14)
    ref = None
    for r in range(5):
        blob = torch.full((pb,), (0x33 * (r + 1)) & 0xFF, dtype=torch.uint8, device=DEV)
        assert L.trw_prepare_csr_hip(d_rp.data_ptr(), n, d_ci.data_ptr(), nnz, 14, blob.data_ptr(), pb, stream) == 0
        torch.cuda.synchronize()
        ext = blob[off.value:off.value + size.value].view(torch.int32).view(nnz, 8).clone()
        if ref is None:
            ref = ext
        else:
            bad = (ext != ref).any(1).nonzero().flatten()
            assert bad.numel() == 0, (r, bad[:5].tolist(), ext[bad[:2]].tolist(), ref[bad[:2]].tolist())


def test_tile_pass_at_every_boundary_of_its_layout(rw):
    """Rows of exactly the lengths at which the marking passes change their ways -- 8 | 9 (the owning thread / the tile pass),
    32 | 33, 63..65 (one task of 64 entries / two), 127..129 (exact mask / filter), 255..257 (the one-wave kernel / the
    workgroup's; the first cap), 1 023..1 025 (the second cap), 2 046..2 049 (a row that fits the 2 048-entry tile whole / one
    that keeps every second entry in it), 4 096, 8 193 and 20 000 (every third, fifth, tenth) -- each pointing at each: every
    (length of u's row, length of v's row) pair occurs as an entry, in both orders.  The 32 bytes of sampled entries against
    numpy, for both cap levels."""
    from torch_random_walk_amd import _native
    L = _native.cabi()
    rng = np.random.default_rng(11)
    lens = [8, 9, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 1023, 1024, 1025, 2046, 2047, 2048, 2049, 4096, 8193, 20000]
    k = len(lens)
    n = 30000
    rows = []
    for i, ln in enumerate(lens):                       # node i: edges to all k special nodes, the rest from a shared pool
        rest = rng.choice(np.arange(k, n), ln - k, replace=False) if ln > k else np.empty(0, np.int64)
        row = np.concatenate([np.arange(k)[:min(k, ln)], rest])
        if ln > 300:                                    # parallel edges here and there
            row[-3:] = row[-4]
        rows.append(np.sort(row))
    src = [np.full(r.size, i) for i, r in enumerate(rows)] + [np.arange(k, n)]
    dst = rows + [(np.arange(k, n) + 1) % n]            # the other nodes: a ring (no dead ends)
    src, dst = np.concatenate(src), np.concatenate(dst)
    order = np.lexsort((dst, src))
    src, dst = src[order], dst[order]
    rp = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(src, minlength=n), out=rp[1:])
    ci = dst.astype(np.int64)
    nnz = len(ci)
    deg = np.diff(rp)
    assert sorted(deg[:k].tolist()) == sorted(lens)
    d_rp, d_ci = _dev(rp, ci)
    special = np.nonzero((src < k) & (ci < k))[0]       # the k x k entries between the special nodes (those of length < k aside)
    assert special.size >= (k - 2) * k
    for what, cap in ((6, 256), (14, 1024)):
        ext = _prepared_ext32(L, d_rp, d_ci, n, nnz, what)
        for e in special:
            u, v = src[e], ci[e]
            row_u, row_v = ci[rp[u]:rp[u + 1]], ci[rp[v]:rp[v + 1]]
            assert ext[e, 0] == v and ext[e, 1] == rp[v] and (ext[e, 2] & 0x7FFFFFFF) == deg[v]
            at = np.nonzero(row_v == u)[0]
            want_rpos = 0 if at.size == 0 else ((at.size << 28) | int(at[0]) if at.size <= 15 else 1)
            assert ext[e, 3] == want_rpos, (what, u, v, ext[e, 3], want_rpos)
            if min(deg[u], deg[v]) <= cap:
                bits = np.zeros(128, dtype=bool)
                np.logical_or.at(bits, np.nonzero(np.isin(row_v, row_u))[0] % 128, True)
                want = np.packbits(bits, bitorder="little").view(np.uint32)
                assert (ext[e, 4:] == want).all(), (what, int(deg[u]), int(deg[v]), ext[e, 4:], want)
                assert bool(ext[e, 2] >> 31) == (deg[v] > 128)
            else:                                       # beyond the cap: all ones, "ask the edge set"
                assert (ext[e, 4:] == 0xFFFFFFFF).all() and bool(ext[e, 2] >> 31)
    # and the walk over it agrees with the oracle (start nodes among the special ones: hub -> hub steps every time)
    tg = rng.integers(0, k, 1000)
    ref = O.walk(rp, ci, tg, 0.25, 4.0, 8, 5, rng_mode=O.RNG_WALKER)
    rw.clear_cache()
    assert rw.prepare(d_rp, d_ci, marks="full") & rw.PREP_MARKS_FULL
    got = rw.walk_ex(d_rp, d_ci, torch.as_tensor(tg).to(DEV), 0.25, 4.0, 8, 5, rng_mode=rw.RNG_WALKER)
    assert (got.cpu().numpy() == ref).all()


def test_prepared_edge_list_is_cached_per_padding_id_and_invalidated_by_writes(rw):
    from torch_rw import utils
    rng = np.random.default_rng(3)
    n = 500
    src, dst = rng.integers(0, n, 4000), rng.integers(0, n, 4000)
    src = src[src % 7 != 3]
    dst = dst[:src.size]
    index, edges = utils.build_node_edge_index(torch.as_tensor(np.stack([src, dst], 1)), torch.arange(n))
    tg = rng.integers(0, n, 3000)
    d_e, d_i, d_t = edges.to(DEV), index.to(DEV), torch.as_tensor(tg).to(DEV)

    def both(pad):
        got = rw.walk_edge_list_ex(d_e, d_i, d_t, 0.25, 4.0, 25, 5, pad, rng_mode=rw.RNG_WALKER).cpu().numpy()
        ref = O.walk_edge_list(d_e.cpu().numpy(), d_i.cpu().numpy(), tg, 0.25, 4.0, 25, 5, pad, rng_mode=O.RNG_WALKER)
        assert (got == ref).all()

    both(n)
    info = rw.cache_info()
    assert len(info) == 1 and info[0][:2] == (n, edges.shape[0]) and info[0][2] == 3      # table + entries
    blob = info[0][3]
    both(n)
    assert rw.cache_info()[0][3] == blob and len(rw.cache_info()) == 1
    both(3)                                                       # another padding id: another prepared edge list
    assert len(rw.cache_info()) == 2
    # in-place write to the tails: the version counter invalidates, the walk follows the new edges
    perm = torch.randperm(d_e.shape[0], device=DEV)
    d_e[:, 1] = d_e[perm, 1]
    both(n)
    u = rw.walk_edge_list_ex(d_e, d_i, d_t, 1.0, 1.0, 25, 5, n, rng_mode=rw.RNG_WALKER).cpu().numpy()
    assert (u == O.walk_edge_list(d_e.cpu().numpy(), d_i.cpu().numpy(), tg, 1.0, 1.0, 25, 5, n, rng_mode=O.RNG_WALKER)).all()


def test_a_rewritten_graph_leaves_no_stale_entry_behind(rw):
    """an entry whose tensors are alive but were written in place can never match again -- it is dropped
    when the same tensors come back (pinned or not), not kept beside its successor until LRU gets to it."""
    rp, ci = hub_graph(2000, 5)
    rp2, ci2 = random_graph(2000, 4, 5)   # same entry count (5 per node), other edges
    assert len(ci2) == len(ci)
    tg = np.arange(2000)
    d_rp, d_ci, d_tg = _dev(rp, ci, tg)
    rw.walk(d_rp, d_ci, d_tg, 0.25, 4.0, 20, 3)
    assert len(rw.cache_info()) == 1
    d_rp.copy_(torch.as_tensor(rp2))
    d_ci.copy_(torch.as_tensor(ci2))
    got = rw.walk(d_rp, d_ci, d_tg, 0.25, 4.0, 20, 3)
    assert len(rw.cache_info()) == 1
    assert (got.cpu().numpy() == O.walk(rp2, ci2, tg, 0.25, 4.0, 20, 3, rng_mode=O.RNG_WALKER)).all()
    # prepare() -> write -> prepare(): the pinned stale blob goes too
    rw.prepare(d_rp, d_ci, marks=True)
    assert len(rw.cache_info()) == 1
    d_ci.copy_(torch.as_tensor(ci2))          # (same values: the version counter moves all the same)
    rw.prepare(d_rp, d_ci, marks=True)
    assert len(rw.cache_info()) == 1
    # a write torch cannot see (through .data) leaves the entry in place: rw.invalidate() is the way to say so
    d_ci.data.copy_(torch.as_tensor(ci))
    d_rp.data.copy_(torch.as_tensor(rp))
    assert rw.invalidate(d_rp, d_ci) == 1 and rw.cache_info() == []
    got = rw.walk(d_rp, d_ci, d_tg, 0.25, 4.0, 20, 3)
    assert (got.cpu().numpy() == O.walk(rp, ci, tg, 0.25, 4.0, 20, 3, rng_mode=O.RNG_WALKER)).all()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs in one process")
def test_two_devices_in_one_process(rw):
    """One process, two cards: each device keeps its own prepared graphs (capacity counts per device) and the deferred
    status read-back of a call on cuda:1 uses an event of cuda:1 (the slots were shared across devices)."""
    rp, ci = hub_graph(2000, 5)
    tg = np.arange(2000)
    ref = O.walk(rp, ci, tg, 0.25, 4.0, 20, 3, rng_mode=O.RNG_WALKER)
    graphs = []
    for dev in ("cuda:0", "cuda:1"):
        for k in range(2):
            t = [torch.as_tensor(np.asarray(a), dtype=torch.int64).to(dev) for a in (rp, ci, tg)]
            graphs.append(t)
            for _ in range(3):                       # deferred checks: slots freed on one device, reused on the other
                got = rw.walk(t[0], t[1], t[2], 0.25, 4.0, 20, 3)
            assert (got.cpu().numpy() == ref).all()
    assert len(rw.cache_info()) == 4                 # two graphs on each of the two devices
    rw.check_status(wait=True)
