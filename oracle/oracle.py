"""ctypes front-end of the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg import this module.  The product package (``torch_random_walk_amd``) never
does; it fails loudly when its HIP library is missing instead.

The C source (``trw_oracle.c``) restates the reference's
``csrc/cpu/rw_cpu.cpp`` and cites it function by function.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libtrw_oracle.so")

RNG_GLIBC_SEQ = 0
RNG_WALKER = 1
RNG_LIBC = 2
RNG_GLIBC_INDEXED = 3

E_ARG, E_DEGREE0, E_RANGE = -1, -2, -3


class Counters(ctypes.Structure):
    _fields_ = [
        ("draws_total", ctypes.c_uint64),
        ("trials_total", ctypes.c_uint64),
        ("membership_tests", ctypes.c_uint64),
        ("scan_entries_total", ctypes.c_uint64),
    ]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


def build(force=False):
    """Compile libtrw_oracle.so with gcc (a few hundred ms)."""
    src = os.path.join(_HERE, "trw_oracle.c")
    hdr = os.path.join(_HERE, "trw_oracle.h")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _LIB_PATH
    subprocess.check_call(
        # same flags as oracle/Makefile (gnu11 + pthread: the multi-threaded timing leg uses pthreads/clock_gettime)
        ["gcc", "-O2", "-fPIC", "-Wall", "-Wextra", "-std=gnu11", "-pthread", "-shared", "-o", _LIB_PATH, src, "-lm"],
        cwd=_HERE)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        i64p = ctypes.POINTER(ctypes.c_int64)
        i32p = ctypes.POINTER(ctypes.c_int32)
        u32p = ctypes.POINTER(ctypes.c_uint32)
        L.trw_oracle_glibc_stream.argtypes = [ctypes.c_int32, ctypes.c_uint64, ctypes.c_uint64, i32p]
        L.trw_oracle_glibc_stream.restype = None
        L.trw_oracle_libc_stream.argtypes = [ctypes.c_int32, ctypes.c_uint64, i32p]
        L.trw_oracle_libc_stream.restype = None
        L.trw_oracle_glibc_window_at.argtypes = [ctypes.c_int32, ctypes.c_uint64, u32p]
        L.trw_oracle_glibc_window_at.restype = None
        L.trw_oracle_philox4x32_10.argtypes = [u32p, u32p, u32p]
        L.trw_oracle_philox4x32_10.restype = None
        L.trw_oracle_walker_draw.argtypes = [ctypes.c_int32, ctypes.c_uint64, ctypes.c_uint64]
        L.trw_oracle_walker_draw.restype = ctypes.c_int32
        L.trw_oracle_probs.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.POINTER(ctypes.c_double)]
        L.trw_oracle_probs.restype = None
        L.trw_oracle_accept_threshold.argtypes = [ctypes.c_double]
        L.trw_oracle_accept_threshold.restype = ctypes.c_uint32
        L.trw_oracle_walk_csr.argtypes = [
            i64p, ctypes.c_int64, i64p, ctypes.c_int64, i64p, ctypes.c_int64, ctypes.c_int64,
            ctypes.c_double, ctypes.c_double, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
            i64p, ctypes.POINTER(Counters), i64p]
        L.trw_oracle_walk_csr.restype = ctypes.c_int
        L.trw_oracle_walk_triples.argtypes = [
            i64p, ctypes.c_int64, i64p, ctypes.c_int64, i64p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32,
            ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, i64p, ctypes.POINTER(ctypes.c_uint64)]
        L.trw_oracle_walk_triples.restype = ctypes.c_int
        L.trw_oracle_walk_edge_list.argtypes = [
            i64p, ctypes.c_int64, i64p, ctypes.c_int64, i64p, ctypes.c_int64, ctypes.c_int64, ctypes.c_double,
            ctypes.c_double, ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, i64p,
            ctypes.POINTER(ctypes.c_uint64)]
        L.trw_oracle_walk_edge_list.restype = ctypes.c_int
        L.trw_oracle_to_windows.argtypes = [i64p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_int64,
                                            ctypes.c_int32, ctypes.c_int32, i64p, i64p, i64p]
        L.trw_oracle_to_windows.restype = ctypes.c_int
        L.trw_oracle_to_windows_cbow.argtypes = [i64p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_int64,
                                                 ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, i64p, i64p, i64p]
        L.trw_oracle_to_windows_cbow.restype = ctypes.c_int
        L.trw_oracle_to_windows_triples.argtypes = [
            i64p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_int64, i64p, ctypes.c_int64, ctypes.c_int32,
            ctypes.c_int32, i64p, i64p, i64p]
        L.trw_oracle_to_windows_triples.restype = ctypes.c_int
        L.trw_oracle_to_windows_triples_cbow.argtypes = [
            i64p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_int64, i64p, ctypes.c_int64, ctypes.c_int32,
            ctypes.c_int32, ctypes.c_int64, i64p, i64p, i64p]
        L.trw_oracle_to_windows_triples_cbow.restype = ctypes.c_int
        _lib = L
    return _lib


def _i64(a):
    a = np.ascontiguousarray(np.asarray(a), dtype=np.int64)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))


def glibc_stream(seed, n, start=0):
    out = np.empty(int(n), dtype=np.int32)
    lib().trw_oracle_glibc_stream(int(seed), int(start), int(n), out.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    return out


def libc_stream(seed, n):
    out = np.empty(int(n), dtype=np.int32)
    lib().trw_oracle_libc_stream(int(seed), int(n), out.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    return out


def glibc_window_at(seed, pos):
    out = np.empty(31, dtype=np.uint32)
    lib().trw_oracle_glibc_window_at(int(seed), int(pos), out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)))
    return out


def philox4x32_10(ctr, key):
    c = np.asarray(ctr, dtype=np.uint32).copy()
    k = np.asarray(key, dtype=np.uint32).copy()
    o = np.empty(4, dtype=np.uint32)
    u32p = ctypes.POINTER(ctypes.c_uint32)
    lib().trw_oracle_philox4x32_10(c.ctypes.data_as(u32p), k.ctypes.data_as(u32p), o.ctypes.data_as(u32p))
    return o


def walker_draw(seed, walker, draw):
    return int(lib().trw_oracle_walker_draw(int(seed), int(walker), int(draw)))


def probs(p, q):
    out = (ctypes.c_double * 3)()
    lib().trw_oracle_probs(float(p), float(q), out)
    return [out[0], out[1], out[2]]


def accept_threshold(prob):
    return int(lib().trw_oracle_accept_threshold(float(prob)))


class OracleError(RuntimeError):
    def __init__(self, code):
        self.code = code
        super().__init__({E_ARG: "bad argument", E_DEGREE0: "walker reached a degree-0 node",
                          E_RANGE: "node id out of range"}.get(code, "oracle error %d" % code))


def walk(row_ptr, col_idx, target_nodes, p, q, walk_length, seed, rng_mode=RNG_GLIBC_SEQ,
         walker_base=0, return_counters=False):
    """The reference's ``walk_cpu`` restated: returns ``[N, walk_length+1]`` int64.

    ``rng_mode=RNG_GLIBC_SEQ`` is the reference's single-thread behaviour
    (bit-exact with csrc/cpu); ``RNG_WALKER`` is the per-walker counter stream.
    """
    rp, rp_p = _i64(row_ptr)
    ci, ci_p = _i64(col_idx)
    tg, tg_p = _i64(target_nodes)
    n = int(tg.shape[0])
    out = np.empty((n, int(walk_length) + 1), dtype=np.int64)
    cnt = Counters()
    dpw = np.zeros(n, dtype=np.int64)
    rc = lib().trw_oracle_walk_csr(
        rp_p, int(rp.shape[0]) - 1, ci_p, int(ci.shape[0]), tg_p, n, int(walker_base),
        float(p), float(q), int(walk_length), int(seed), int(rng_mode),
        out.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), ctypes.byref(cnt),
        dpw.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
    if rc != 0:
        raise OracleError(rc)
    if return_counters:
        d = cnt.as_dict()
        d["draws_per_walker"] = dpw
        return out, d
    return out


def walk_timed(row_ptr, col_idx, target_nodes, p, q, walk_length, seed, rng_mode=RNG_LIBC):
    """Run the walk without counters and return (walks, seconds) -- the CPU baseline leg."""
    import time
    rp, rp_p = _i64(row_ptr)
    ci, ci_p = _i64(col_idx)
    tg, tg_p = _i64(target_nodes)
    n = int(tg.shape[0])
    out = np.empty((n, int(walk_length) + 1), dtype=np.int64)
    t0 = time.perf_counter()
    rc = lib().trw_oracle_walk_csr(
        rp_p, int(rp.shape[0]) - 1, ci_p, int(ci.shape[0]), tg_p, n, 0,
        float(p), float(q), int(walk_length), int(seed), int(rng_mode),
        out.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), None, None)
    dt = time.perf_counter() - t0
    if rc != 0:
        raise OracleError(rc)
    return out, dt


def walk_timed_mt(row_ptr, col_idx, target_nodes, p, q, walk_length, seed, n_threads):
    """TIMING only: the reference's multi-threaded policy (at::parallel_for chunks, every thread on the one
    locked libc rand()).  Returns (walks, seconds, threads the policy starts); the walks depend on the
    thread interleaving and are only good for the edge property."""
    rp, rp_p = _i64(row_ptr)
    ci, ci_p = _i64(col_idx)
    tg, tg_p = _i64(target_nodes)
    n = int(tg.shape[0])
    out = np.empty((n, int(walk_length) + 1), dtype=np.int64)
    sec, used = ctypes.c_double(0.0), ctypes.c_int32(0)
    f = lib().trw_oracle_walk_csr_mt
    f.restype = ctypes.c_int
    rc = f(rp_p, ctypes.c_int64(int(rp.shape[0]) - 1), ci_p, ctypes.c_int64(int(ci.shape[0])), tg_p, ctypes.c_int64(n),
           ctypes.c_double(float(p)), ctypes.c_double(float(q)), ctypes.c_int32(int(walk_length)),
           ctypes.c_int32(int(seed)), ctypes.c_int32(int(n_threads)),
           out.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), ctypes.byref(sec), ctypes.byref(used))
    if rc != 0:
        raise OracleError(rc)
    return out, float(sec.value), int(used.value)


def walk_triples(triples_indexed, relation_tail_index, target_nodes, walk_length, padding_idx, seed,
                 rng_mode=RNG_GLIBC_SEQ, walker_base=0, return_draws=False, timed=False):
    """The reference's ``triples::walk_triples_cpu`` restated: ``[N, 2*walk_length+1]`` int64."""
    import time
    tr = np.ascontiguousarray(np.asarray(triples_indexed), dtype=np.int64).reshape(-1, 3)
    ix = np.ascontiguousarray(np.asarray(relation_tail_index), dtype=np.int64).reshape(-1, 2)
    tg, tg_p = _i64(target_nodes)
    n = int(tg.shape[0])
    out = np.empty((n, 2 * int(walk_length) + 1), dtype=np.int64)
    draws = ctypes.c_uint64(0)
    p64 = ctypes.POINTER(ctypes.c_int64)
    t0 = time.perf_counter()
    rc = lib().trw_oracle_walk_triples(tr.ctypes.data_as(p64), tr.shape[0], ix.ctypes.data_as(p64), ix.shape[0],
                                       tg_p, n, int(walker_base), int(walk_length), int(padding_idx), int(seed),
                                       int(rng_mode), out.ctypes.data_as(p64), ctypes.byref(draws))
    dt = time.perf_counter() - t0
    if rc != 0:
        raise OracleError(rc)
    if timed:
        return out, dt
    if return_draws:
        return out, int(draws.value)
    return out


def walk_edge_list(edge_list_indexed, node_edge_index, target_nodes, p, q, walk_length, seed, padding_idx,
                   restart=True, rng_mode=RNG_GLIBC_SEQ, walker_base=0, return_draws=False):
    """The reference's ``walk_edge_list_cpu`` restated: ``[N, walk_length+1]`` int64."""
    ed = np.ascontiguousarray(np.asarray(edge_list_indexed), dtype=np.int64).reshape(-1, 2)
    ix = np.ascontiguousarray(np.asarray(node_edge_index), dtype=np.int64).reshape(-1, 2)
    tg, tg_p = _i64(target_nodes)
    n = int(tg.shape[0])
    out = np.empty((n, int(walk_length) + 1), dtype=np.int64)
    draws = ctypes.c_uint64(0)
    p64 = ctypes.POINTER(ctypes.c_int64)
    rc = lib().trw_oracle_walk_edge_list(ed.ctypes.data_as(p64), ed.shape[0], ix.ctypes.data_as(p64), ix.shape[0],
                                         tg_p, n, int(walker_base), float(p), float(q), int(walk_length), int(seed),
                                         int(padding_idx), 1 if restart else 0, int(rng_mode),
                                         out.ctypes.data_as(p64), ctypes.byref(draws))
    if rc != 0:
        raise OracleError(rc)
    if return_draws:
        return out, int(draws.value)
    return out


def _windows_common(walks, window_size):
    wk = np.ascontiguousarray(np.asarray(walks), dtype=np.int64)
    assert wk.ndim == 2
    num_walks, cols = wk.shape
    W = (cols - int(window_size) + 1) * num_walks
    return wk, num_walks, cols, max(W, 0)


def to_windows(walks, window_size, num_nodes, seed, rng_mode=RNG_GLIBC_SEQ):
    """The reference's ``to_windows_cpu`` restated: (target_nodes, pos_windows, neg_windows)."""
    wk, num_walks, cols, W = _windows_common(walks, window_size)
    tgt = np.empty(W, dtype=np.int64)
    pos = np.empty((W, int(window_size) - 1), dtype=np.int64)
    neg = np.empty((W, int(window_size) - 1), dtype=np.int64)
    p64 = ctypes.POINTER(ctypes.c_int64)
    rc = lib().trw_oracle_to_windows(wk.ctypes.data_as(p64), num_walks, cols, int(window_size), int(num_nodes),
                                     int(seed), int(rng_mode), tgt.ctypes.data_as(p64), pos.ctypes.data_as(p64),
                                     neg.ctypes.data_as(p64))
    if rc != 0:
        raise OracleError(rc)
    return tgt, pos, neg


def to_windows_cbow(walks, window_size, num_nodes, seed, rng_mode=RNG_GLIBC_SEQ, window_base=0):
    """The reference's ``to_windows_cbow_cpu`` restated: (pos_nodes, neg_nodes, windows)."""
    wk, num_walks, cols, W = _windows_common(walks, window_size)
    posn = np.empty(W, dtype=np.int64)
    negn = np.empty(W, dtype=np.int64)
    win = np.empty((W, int(window_size) - 1), dtype=np.int64)
    p64 = ctypes.POINTER(ctypes.c_int64)
    rc = lib().trw_oracle_to_windows_cbow(wk.ctypes.data_as(p64), num_walks, cols, int(window_size), int(num_nodes),
                                          int(seed), int(rng_mode), int(window_base), posn.ctypes.data_as(p64),
                                          negn.ctypes.data_as(p64), win.ctypes.data_as(p64))
    if rc != 0:
        raise OracleError(rc)
    return posn, negn, win


def _triples_windows_common(walks, triples, window_size):
    wk = np.ascontiguousarray(np.asarray(walks), dtype=np.int64)
    tr = np.ascontiguousarray(np.asarray(triples), dtype=np.int64).reshape(-1, 3)
    num_walks, cols = wk.shape
    W = ((cols - 1) // 2) * num_walks
    return wk, tr, num_walks, cols, W


def to_windows_triples(walks, window_size, num_nodes, padding_idx, triples, seed, rng_mode=RNG_GLIBC_SEQ):
    """``to_windows_triples_cpu`` restated: (target_triples [W,3], pos_windows [W,2ws,3], neg_windows [W,2ws,3])."""
    wk, tr, num_walks, cols, W = _triples_windows_common(walks, triples, window_size)
    ws = int(window_size)
    tgt = np.empty((W, 3), dtype=np.int64)
    pos = np.empty((W, 2 * ws, 3), dtype=np.int64)
    neg = np.empty((W, 2 * ws, 3), dtype=np.int64)
    p64 = ctypes.POINTER(ctypes.c_int64)
    rc = lib().trw_oracle_to_windows_triples(wk.ctypes.data_as(p64), num_walks, cols, ws, int(padding_idx),
                                             tr.ctypes.data_as(p64), tr.shape[0], int(seed), int(rng_mode),
                                             tgt.ctypes.data_as(p64), pos.ctypes.data_as(p64), neg.ctypes.data_as(p64))
    if rc != 0:
        raise OracleError(rc)
    return tgt, pos, neg


def to_windows_triples_cbow(walks, window_size, num_nodes, padding_idx, triples, seed, rng_mode=RNG_GLIBC_SEQ,
                            window_base=0):
    """``to_windows_triples_cbow_cpu`` restated: (pos_triples [W,3], neg_triples [W,3], pos_windows [W,2ws,3])."""
    wk, tr, num_walks, cols, W = _triples_windows_common(walks, triples, window_size)
    ws = int(window_size)
    post = np.empty((W, 3), dtype=np.int64)
    negt = np.empty((W, 3), dtype=np.int64)
    pos = np.empty((W, 2 * ws, 3), dtype=np.int64)
    p64 = ctypes.POINTER(ctypes.c_int64)
    rc = lib().trw_oracle_to_windows_triples_cbow(wk.ctypes.data_as(p64), num_walks, cols, ws, int(padding_idx),
                                                  tr.ctypes.data_as(p64), tr.shape[0], int(seed), int(rng_mode),
                                                  int(window_base), post.ctypes.data_as(p64),
                                                  negt.ctypes.data_as(p64), pos.ctypes.data_as(p64))
    if rc != 0:
        raise OracleError(rc)
    return post, negt, pos
