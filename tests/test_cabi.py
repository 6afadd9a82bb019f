"""The C ABI library: loads, exports what include/trw.h declares, host helpers agree with the oracle.
No kernel is launched here (no GPU in this tier)."""
import ctypes
import os
import re

import numpy as np
import pytest

from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "trw.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(trw_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    from torch_random_walk_amd import _native
    lib = ctypes.CDLL(_native.HIP_LIB_PATH)
    names = _declared_functions()
    assert "trw_walk_csr_hip" in names and len(names) >= 8
    for name in names:
        assert hasattr(lib, name), "libtrw_hip.so does not export %s" % name


def test_version_and_constants():
    from torch_random_walk_amd import _native
    assert _native.cabi().trw_version() == 100
    assert (_native.RNG_AUTO, _native.RNG_GLIBC, _native.RNG_WALKER) == (0, 1, 2)
    assert _native.ext.abi_version() == 100


@pytest.mark.parametrize("p,q", [(0.25, 4.0), (0.7, 0.5), (2.0, 1.0), (1.0, 1.0 + 1e-12), (1e-3, 1e3), (5.0, 0.2)])
def test_accept_threshold_matches_oracle(p, q):
    from torch_random_walk_amd import _native
    for prob in O.probs(p, q):
        assert _native.cabi().trw_accept_threshold(prob) == O.accept_threshold(prob)


@pytest.mark.parametrize("seed", [0, 1, 10, -1, 2**31 - 1, -2**31])
@pytest.mark.parametrize("pos", [0, 1, 255, 256, 80 * 12345, 2**24 + 17, 2**40 + 3, 2**48 - 1])
def test_jump_tables_match_oracle(seed, pos):
    from torch_random_walk_amd import _native
    w = (ctypes.c_uint32 * 31)()
    _native.cabi().trw_glibc_window_at(seed, pos, w)
    assert list(w) == O.glibc_window_at(seed, pos).tolist()


def test_window_reproduces_libc_draws(libc_vectors):
    """Stepping the recurrence from a jumped window gives the real libc outputs."""
    from torch_random_walk_amd import _native
    deep = libc_vectors["deep"]
    samples = deep["samples"]["10"]
    for k in (1, 7, len(samples) - 1):
        pos = k * deep["stride"]
        w = (ctypes.c_uint32 * 31)()
        _native.cabi().trw_glibc_window_at(10, pos, w)
        draw = ((w[0] + w[28]) & 0xFFFFFFFF) >> 1
        assert draw == samples[k]


def test_workspace_bytes():
    """One-shot workspace = the call's own scratch (status block, RNG windows) + the prepared-graph blob
    (256-byte header, uint32 shadow, and by call shape the fat entries / the edge table)."""
    from torch_random_walk_amd import _native
    lib = _native.cabi()
    f = lib.trw_walk_csr_workspace_bytes
    n, nnz, N, L = 10, 20, 1000, 80
    head = 256 + 256                                                  # status block + prepared header
    shadow = 4 * (n + 1) + 4 * nnz
    stage = 4 * N * (L + 1)
    assert f(n, nnz, 0, 5, 1.0, 1.0, 0) >= 256
    assert f(1000, 10**8, 5, 6, 0.5, 2.0, 1) == head                  # short serial glibc replay: int64 kernels, no shadow
    assert f(n, nnz, N, L, 1.0, 1.0, 2) == head                       # uniform on the walker stream: int64 kernels
    table = 32 * 256                                                  # edge set: 2.5 places of 8 bytes per edge in blocks of 4, >= 256 blocks
    fat = 32 * nnz                                                    # 32-byte entries of the biased walk
    # a node2vec call worth the entries gets the masks of rows up to 256 entries with them (the tile pass makes them
    # cheaper to build than doing without): + the marking passes' scratch (items for every entry, task records, two
    # words per node)
    scratch = lib.trw_prepare_csr_scratch_bytes(n, nnz, 2 | 4)
    assert 8 * nnz < scratch < 8 * nnz + 8 * (n + 1) + 2 * 8 * 4096 + 12 * 256
    assert shadow + table + fat + scratch <= f(n, nnz, 1, L, 0.5, 2.0, 2) - head <= shadow + table + fat + scratch + 6 * 256   # 4 steps per entry
    uniform = f(n, nnz, N, L, 1.0, 1.0, 0)                            # windows + shadow; rows written in place
    assert head + N * 124 + shadow <= uniform < head + N * 124 + shadow + stage
    assert f(2**31, nnz, N, L, 0.5, 2.0, 2) == head                   # too many nodes for the uint32 shadow
    assert f(n, 2**32, N, L, 0.5, 2.0, 2) == head
    # the prepared blob by itself, and what a call over it needs
    g = lib.trw_prepare_csr_bytes
    assert g(n, nnz, 0) >= 256 + shadow and g(n, nnz, 1) >= g(n, nnz, 0) + 16 * nnz
    assert g(n, nnz, 2) >= g(n, nnz, 0) + table + fat and g(n, nnz, 2 | 4) > g(n, nnz, 2)
    assert g(n, nnz, -1) == 256 and g(n, nnz, -2) == 256
    w = lib.trw_walk_csr_prepared_workspace_bytes
    assert w(N, L, 0.5, 2.0, 2) == 256 and 256 + N * 124 <= w(N, L, 1.0, 1.0, 0) < 256 + N * 124 + 1024
    plan = lib.trw_prepare_plan
    assert plan(n, nnz, 1, L, 0.5, 2.0, 2) == 2 | 4 and plan(1000, 10**6, 100, L, 0.5, 2.0, 2) == 0
    # the masks of rows up to 256 entries come with the entries; one call that makes 9 steps per entry brings those of
    # rows up to 1 024 along
    assert plan(n, nnz, 2, L, 0.5, 2.0, 2) == 2 | 4 and plan(n, nnz, 3, L, 0.5, 2.0, 2) == 2 | 4 | 8
    assert plan(n, nnz, N, L, 0.5, 2.0, 2) == 2 | 4 | 8
    assert plan(1000, 10**8, 5, 6, 0.5, 2.0, 1) == -1 and plan(n, nnz, N, L, 1.0, 1.0, 2) == -2
    assert plan(n, nnz, N, L, 0.5, 2.0, 1) == 2 | 4      # a serial replay of this many steps rides the prepared graph
    # a graph the size of BASELINE configs[3]: 20 bytes of edge set + 32 of entry + 4 of shadow per edge (it was a
    # 2^32-slot, 69 GB table until round 4)
    big = g(111_000_000, 1_711_000_000, 2)
    assert 56 * 1_711_000_000 < big < 56 * 1_711_000_000 + 10**9
    # with every mask built: what the walks read stays 56 bytes per edge + 4 per node; the marking passes' scratch (items
    # for EVERY entry, the worst case, and the task records) is apart from it and can be handed in separately
    keep, scratch = lib.trw_prepare_csr_persistent_bytes, lib.trw_prepare_csr_scratch_bytes
    for nn, ee in ((111_000_000, 1_711_000_000), (10_000_000, 110_000_000)):
        assert keep(nn, ee, 2 | 4 | 8) == keep(nn, ee, 2) == g(nn, ee, 2) and scratch(nn, ee, 2) == 0
        assert g(nn, ee, 2 | 4 | 8) == keep(nn, ee, 2 | 4 | 8) + scratch(nn, ee, 2 | 4 | 8)
        assert 8 * ee < scratch(nn, ee, 2 | 4 | 8) < 11 * ee + 8 * nn + 10**6
        assert keep(nn, ee, 2 | 4 | 8) < 57 * ee + 4 * nn


def test_prepared_graph_regions_are_disjoint_and_inside_the_blob():
    """trw_prepare_csr_region (a host function: the layout depends on the shapes alone): the parts of a prepared graph
    lie inside trw_prepare_csr_bytes, in order, without overlap; parts a `what` does not have report 0 bytes."""
    import ctypes
    from torch_random_walk_amd import _native
    lib = _native.cabi()
    for n, nnz in ((10, 20), (100_000, 1_100_000), (10_000_000, 110_000_000), (111_000_000, 1_711_000_000)):
        for what in (0, 1, 2, 3, 6, 14, 15):
            total = lib.trw_prepare_csr_bytes(n, nnz, what)
            spans = {}
            for which in range(7):
                off, size = ctypes.c_size_t(), ctypes.c_size_t()
                assert lib.trw_prepare_csr_region(n, nnz, what, which, ctypes.byref(off), ctypes.byref(size)) == 0
                spans[which] = (off.value, size.value)
                assert off.value + size.value <= total
            assert spans[0] == (0, 256) and spans[1][1] == 4 * (n + 1) and spans[2][1] == 4 * nnz
            assert (spans[3][1] == 16 * nnz) == bool(what & 1)
            assert (spans[5][1] == 32 * nnz) == bool(what & 14) and (spans[4][1] > 0) == bool(what & 14)
            assert (spans[6][1] > 0) == bool(what & 12)
            if what & 12:                                           # the task records: scratch, behind what the walks read
                assert spans[6][0] >= lib.trw_prepare_csr_persistent_bytes(n, nnz, what)
            live = sorted(v for v in spans.values() if v[1])
            assert all(a[0] + a[1] <= b[0] for a, b in zip(live, live[1:])), (n, nnz, what, live)
            if what & 14:
                assert 16 * nnz <= spans[4][1] <= 24 * nnz + 8192        # the edge set: 2.5 places of 8 bytes per edge
    off, size = ctypes.c_size_t(), ctypes.c_size_t()
    assert lib.trw_prepare_csr_region(10, 20, 2, 99, ctypes.byref(off), ctypes.byref(size)) == -1


def test_argument_errors_return_codes_without_touching_the_gpu():
    from torch_random_walk_amd import _native
    L = _native.cabi()
    rc = L.trw_walk_csr_hip(None, 1, None, 0, None, 0, 0, 1.0, 1.0, 3, 1, 0, None, None, 0, None)
    assert rc == -1 and b"row_ptr" in L.trw_last_error()
    buf = np.zeros(8, dtype=np.int64)
    ptr = buf.ctypes.data
    assert L.trw_walk_csr_hip(ptr, 1, ptr, 1, ptr, 1, 0, 1.0, 1.0, 0, 1, 0, ptr, None, 0, None) == -1   # walk_length 0
    assert L.trw_walk_csr_hip(ptr, 1, ptr, 1, ptr, 1, 0, 0.0, 1.0, 3, 1, 0, ptr, None, 0, None) == -1   # p = 0
    assert L.trw_walk_csr_hip(ptr, 1, ptr, 1, ptr, 1, 0, 1.0, 1.0, 3, 1, 0, ptr, None, 0, None) == -2   # no workspace
    assert L.trw_walk_csr_hip(ptr, 1, ptr, 1, ptr, 1, 0, 1.0, 1.0, 3, 1, 7, ptr, ptr, 4096, None) == -4  # rng mode
    assert L.trw_walk_csr_hip(ptr, 1, ptr, 1, ptr, 1, 5, 0.5, 2.0, 3, 1, 1, ptr, ptr, 4096, None) == -4  # serial chain + base
