"""The CPU oracle against the reference's own golden vectors, the real libc and itself.

These pin the oracle; the GPU parity tests then compare the HIP
path with the oracle.
"""
import ctypes

import numpy as np
import pytest

from oracle import oracle as O
from tests.helpers import hub_graph, random_graph


def _graph(golden):
    g = golden["graph"]
    return g["row_ptr"], g["col_idx"], g["nodes"]


# ---- the reference's golden vectors (tests/test_rw.py:49-53, :116-120) ----------------------

@pytest.mark.parametrize("mode", [O.RNG_GLIBC_SEQ, O.RNG_LIBC, O.RNG_GLIBC_INDEXED])
def test_reference_uniform_golden(golden, mode):
    rp, ci, nodes = _graph(golden)
    c = golden["uniform_cpu"]
    w = O.walk(rp, ci, nodes, c["p"], c["q"], c["walk_length"], c["seed"], rng_mode=mode)
    assert w.tolist() == c["walks"]


@pytest.mark.parametrize("mode", [O.RNG_GLIBC_SEQ, O.RNG_LIBC])
def test_reference_biased_golden(golden, mode):
    rp, ci, nodes = _graph(golden)
    c = golden["biased_cpu"]
    w = O.walk(rp, ci, nodes, c["p"], c["q"], c["walk_length"], c["seed"], rng_mode=mode)
    assert w.tolist() == c["walks"]


# ---- glibc rand() restatement ------------------------------------------------------------

def test_glibc_stream_matches_fixture(libc_vectors):
    for seed, vals in libc_vectors["first64"].items():
        assert O.glibc_stream(int(seed), 64).tolist() == vals
    deep = libc_vectors["deep"]
    for seed, samples in deep["samples"].items():
        full = O.glibc_stream(int(seed), deep["n"])
        assert full[:: deep["stride"]].tolist() == samples


@pytest.mark.parametrize("seed", [0, 1, 10, 20, 123456, -1, -12345, -2**31, 2**31 - 1])
def test_glibc_stream_matches_live_libc(seed):
    assert (O.glibc_stream(seed, 5000) == O.libc_stream(seed, 5000)).all()


def test_seed_zero_is_seed_one():
    assert (O.glibc_stream(0, 100) == O.glibc_stream(1, 100)).all()


@pytest.mark.parametrize("start", [1, 2, 30, 31, 32, 343, 344, 5000, 123457, 2**33 + 12345])
def test_jump_ahead_matches_stepping(start):
    if start < 200000:
        full = O.glibc_stream(10, start + 50)
        assert (O.glibc_stream(10, 50, start=start) == full[start:]).all()
    else:  # composition: jump(a+b) == jump(a) then step b
        a = O.glibc_stream(10, 100, start=start)
        b = O.glibc_stream(10, 40, start=start + 60)
        assert (a[60:] == b).all()


# ---- Philox known answers (Random123 kat_vectors) -----------------------------------------

def test_philox_known_answers():
    assert O.philox4x32_10([0, 0, 0, 0], [0, 0]).tolist() == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert O.philox4x32_10([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2).tolist() == \
        [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert O.philox4x32_10([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]).tolist() == \
        [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_walker_draw_layout():
    blk = O.philox4x32_10([7, 0, 3, 0], [10, 0])
    assert [O.walker_draw(10, 7, 12 + k) for k in range(4)] == [int(x) >> 1 for x in blk]
    hi = O.philox4x32_10([5, 2, 0, 0], [0xFFFFFFFF, 0])
    assert O.walker_draw(-1, (2 << 32) + 5, 1) == int(hi[1]) >> 1


# ---- accept threshold == the reference's fp64 compare --------------------------------------

@pytest.mark.parametrize("p,q", [(0.25, 4.0), (0.7, 0.5), (2.0, 1.0), (1.0, 1.0 + 1e-12), (3.0, 7.0), (1e-3, 1e3)])
def test_accept_threshold_is_the_fp64_compare(p, q):
    for prob in O.probs(p, q):
        t = O.accept_threshold(prob)
        for r in range(max(0, t - 3), min(2**31 - 1, t + 3) + 1):
            assert ((r / 2147483647.0) < prob) == (r < t)


def test_probs_follow_reference_formula():
    # rw_cpu.cpp:130-134
    assert O.probs(0.25, 4.0) == [1.0, 0.25, 0.0625]
    assert O.probs(1.0, 1.0) == [1.0, 1.0, 1.0]
    p0, p1, p2 = O.probs(0.7, 0.5)
    assert p2 == 1.0 and p1 == 0.5 and abs(p0 - (1 / 0.7) / 2.0) < 1e-15


# ---- walker semantics -------------------------------------------------------------------

def test_indexed_equals_sequential_uniform():
    rp, ci = random_graph(200, 4, 3)
    tg = np.random.default_rng(0).integers(0, 200, 300)
    for seed in (0, 1, 10, 123456):
        a = O.walk(rp, ci, tg, 1.0, 1.0, 20, seed, rng_mode=O.RNG_GLIBC_SEQ)
        b = O.walk(rp, ci, tg, 1.0, 1.0, 20, seed, rng_mode=O.RNG_GLIBC_INDEXED)
        c = O.walk(rp, ci, tg, 1.0, 1.0, 20, seed, rng_mode=O.RNG_LIBC)
        assert (a == b).all() and (a == c).all()


@pytest.mark.parametrize("p,q", [(0.25, 4.0), (0.7, 0.5), (2.0, 1.0), (1.0, 1.0 + 1e-12)])
def test_biased_sequential_equals_live_libc(p, q):
    rp, ci = random_graph(200, 4, 5)
    tg = np.random.default_rng(1).integers(0, 200, 300)
    a, ca = O.walk(rp, ci, tg, p, q, 20, 10, rng_mode=O.RNG_GLIBC_SEQ, return_counters=True)
    b, cb = O.walk(rp, ci, tg, p, q, 20, 10, rng_mode=O.RNG_LIBC, return_counters=True)
    assert (a == b).all()
    assert ca["draws_total"] == cb["draws_total"] == 300 * 1 + 2 * ca["trials_total"]
    assert (ca["draws_per_walker"] >= 1 + 2 * 19).all()


def test_uniform_draw_count_and_columns():
    rp, ci = random_graph(50, 3, 7)
    tg = np.arange(50)
    w, c = O.walk(rp, ci, tg, 1.0, 1.0, 9, 4, return_counters=True)
    assert w.shape == (50, 10) and (w[:, 0] == tg).all()
    assert c["draws_total"] == 50 * 9 and (c["draws_per_walker"] == 9).all()


def test_exact_equality_dispatch():
    # p = q = 1 + eps takes the biased path (rw_cpu.cpp:220) and walks differently
    rp, ci = random_graph(100, 4, 9)
    tg = np.arange(100)
    a = O.walk(rp, ci, tg, 1.0, 1.0, 10, 3)
    b = O.walk(rp, ci, tg, 1.0, 1.0 + 1e-12, 10, 3)
    assert not (a == b).all()


def test_walker_mode_is_shard_invariant():
    rp, ci = hub_graph(300, 2)
    tg = np.random.default_rng(2).integers(0, 300, 257)
    for p, q in ((1.0, 1.0), (0.25, 4.0)):
        full = O.walk(rp, ci, tg, p, q, 15, 99, rng_mode=O.RNG_WALKER)
        parts = [O.walk(rp, ci, tg[lo:hi], p, q, 15, 99, rng_mode=O.RNG_WALKER, walker_base=lo)
                 for lo, hi in ((0, 100), (100, 101), (101, 257))]
        assert (np.concatenate(parts) == full).all()


def test_uniform_glibc_shards_by_walker_base():
    rp, ci = random_graph(120, 4, 11)
    tg = np.random.default_rng(3).integers(0, 120, 90)
    full = O.walk(rp, ci, tg, 1.0, 1.0, 17, 10)
    for mode in (O.RNG_GLIBC_SEQ, O.RNG_GLIBC_INDEXED):
        part = O.walk(rp, ci, tg[40:], 1.0, 1.0, 17, 10, rng_mode=mode, walker_base=40)
        assert (part == full[40:]).all()


def test_edge_cases():
    rp, ci = random_graph(30, 3, 13)
    # walk_length = 1
    w = O.walk(rp, ci, [3, 4], 0.5, 2.0, 1, 1)
    assert w.shape == (2, 2)
    # empty batch
    assert O.walk(rp, ci, [], 1.0, 1.0, 5, 1).shape == (0, 6)
    # duplicates are independent walkers
    w = O.walk(rp, ci, [7, 7, 7, 7], 1.0, 1.0, 12, 5)
    assert len({tuple(r) for r in w.tolist()}) > 1
    # walk_length < 1 is an argument error (the reference writes out of bounds there)
    with pytest.raises(O.OracleError):
        O.walk(rp, ci, [1], 1.0, 1.0, 0, 1)


def test_degree_zero_is_reported_not_sigfpe():
    rp = np.array([0, 1, 1, 2], dtype=np.int64)  # node 1 has no out-edges
    ci = np.array([1, 0], dtype=np.int64)
    with pytest.raises(O.OracleError) as e:
        O.walk(rp, ci, [0], 1.0, 1.0, 3, 1)
    assert e.value.code == O.E_DEGREE0


def test_is_neighbor_direction():
    # is_neighbor(x, t) scans t's row for x (rw_cpu.cpp:32-56): edge t->x, not x->t.
    # 0->1, 1->2, 1->0, 2->0, 0->2: from walk (0,1) candidate 2: t=0 has edge 0->2 => distance 1.
    rp = np.array([0, 2, 4, 5], dtype=np.int64)
    ci = np.array([1, 2, 0, 2, 0], dtype=np.int64)
    w, c = O.walk(rp, ci, [0] * 64, 0.5, 2.0, 6, 3, return_counters=True)
    assert c["membership_tests"] > 0 and c["scan_entries_total"] >= c["membership_tests"]
    # every step is an edge
    edges = {(0, 1), (0, 2), (1, 0), (1, 2), (2, 0)}
    for row in w.tolist():
        assert all((a, b) in edges for a, b in zip(row, row[1:]))


def test_bias_statistics_match_node2vec():
    """Second-step distribution under the walker stream vs the analytic node2vec weights."""
    # star-with-triangle: t=0, prev=1; N(1) = {0, 2, 3}; 2 is adjacent to 0 (d=1), 3 is not (d=2)
    edges = [(0, 1), (1, 0), (1, 2), (1, 3), (2, 0), (0, 2), (2, 1), (3, 1), (3, 3)]
    n = 4
    src = np.array([e[0] for e in edges])
    dst = np.array([e[1] for e in edges])
    order = np.lexsort((dst, src))
    rp = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(src, minlength=n), out=rp[1:])
    ci = dst[order].astype(np.int64)
    p, q = 0.5, 2.0
    N = 200000
    w = O.walk(rp, ci, [0] * N, p, q, 2, 1234, rng_mode=O.RNG_WALKER)
    sel = w[w[:, 1] == 1]
    counts = np.array([(sel[:, 2] == k).sum() for k in (0, 2, 3)], dtype=np.float64)
    weights = np.array([1 / p, 1.0, 1 / q])
    expect = weights / weights.sum() * counts.sum()
    chi2 = ((counts - expect) ** 2 / expect).sum()
    assert chi2 < 20.0, (counts, expect)


# ---- an independent pure-Python reading of rw_cpu.cpp, fed by the oracle's draw stream ---------------

def _py_walk(row_ptr, col_idx, targets, p, q, L, draws):
    """Small-case restatement in plain Python (rw_cpu.cpp:58-201), consuming `draws` in order."""
    it = iter(draws.tolist())
    RAND_MAX = 2147483647
    uniform = (p == 1.0 and q == 1.0)
    max_prob = max(max(1.0 / p, 1), 1.0 / q)
    p0, p1, p2 = 1.0 / p / max_prob, 1.0 / max_prob, 1.0 / q / max_prob

    def sample(v):
        a, b = row_ptr[v], row_ptr[v + 1]
        return col_idx[a + next(it) % (b - a)]

    out = []
    for tnode in targets:
        walk = [int(tnode)]
        if uniform:
            for _ in range(L):
                walk.append(int(sample(walk[-1])))
        else:
            walk.append(int(sample(walk[0])))
            for s in range(2, L + 1):
                t = walk[s - 2]
                while True:
                    x = int(sample(walk[s - 1]))
                    u = next(it) / RAND_MAX
                    if x == t:
                        if u < p0:
                            break
                    elif x in col_idx[row_ptr[t]:row_ptr[t + 1]]:
                        if u < p1:
                            break
                    elif u < p2:
                        break
                walk.append(x)
        out.append(walk)
    return out


@pytest.mark.parametrize("p,q", [(1.0, 1.0), (0.25, 4.0), (0.7, 0.5), (3.0, 1.0)])
def test_c_oracle_equals_plain_python_restatement(p, q):
    rp, ci = random_graph(40, 3, 17)
    tg = np.random.default_rng(4).integers(0, 40, 25)
    for seed in (1, 10):
        w, c = O.walk(rp, ci, tg, p, q, 9, seed, return_counters=True)
        py = _py_walk(rp.tolist(), ci.tolist(), tg, p, q, 9, O.libc_stream(seed, c["draws_total"] + 8))
        assert w.tolist() == py


def test_multithreaded_timing_leg():
    """trw_oracle_walk_csr_mt: one thread = the sequential libc walk; several threads = the reference's
    at::parallel_for policy on one locked rand() -- interleaving-dependent walks that still follow edges."""
    from tests.helpers import random_graph
    rp, ci = random_graph(3000, 5, 4)
    tg = np.arange(3000).repeat(30)                                   # 90 000 walkers: 3 chunks of the uniform grain
    ref = O.walk(rp, ci, tg, 1.0, 1.0, 10, 7, rng_mode=O.RNG_LIBC)
    w, sec, used = O.walk_timed_mt(rp, ci, tg, 1.0, 1.0, 10, 7, 1)
    assert used == 1 and sec > 0 and (w == ref).all()
    w, sec, used = O.walk_timed_mt(rp, ci, tg, 1.0, 1.0, 10, 7, 4)
    assert used == 3 and (w[:, 0] == tg).all()
    edges = set(zip(np.repeat(np.arange(3000), np.diff(rp)).tolist(), ci.tolist()))
    for row in w[::997]:
        assert all((a, b) in edges for a, b in zip(row[:-1].tolist(), row[1:].tolist()))
    wb, _, used = O.walk_timed_mt(rp, ci, tg[:5000], 0.5, 2.0, 10, 7, 4)     # biased grain: 32768 // 11
    assert used == 2 and (wb[:, 0] == tg[:5000]).all()


def test_oracle_regression_vectors():
    """The oracle still produces, walk for walk, what it produced when its pins were last green
    (tests/golden/make_oracle_regression.py)."""
    import importlib.util
    import json
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_oracle_regression",
                                                  os.path.join(here, "golden", "make_oracle_regression.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    want = json.load(open(os.path.join(here, "golden", "oracle_regression.json")))
    got = {name: mod.digest(fn()) for name, fn in mod.cases()}
    assert got == want
