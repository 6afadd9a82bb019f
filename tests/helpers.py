"""Shared small-graph builders for the tests (numpy, seeded)."""
import numpy as np


def random_graph(n, avg_deg, seed, sort_rows=True, ring=True):
    """Directed random graph as int64 CSR numpy arrays, min out-degree >= 1 via a ring."""
    rng = np.random.default_rng(seed)
    m = n * avg_deg
    src = rng.integers(0, n, m)
    dst = rng.integers(0, n, m)
    if ring:
        src = np.concatenate([src, np.arange(n)])
        dst = np.concatenate([dst, (np.arange(n) + 1) % n])
    order = np.lexsort((dst, src)) if sort_rows else np.argsort(src, kind="stable")
    src, dst = src[order], dst[order]
    row_ptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(src, minlength=n), out=row_ptr[1:])
    return row_ptr, dst.astype(np.int64)


def hub_graph(n, seed):
    """Power-law-ish graph: node 0 is a hub adjacent (both ways) to ~half the nodes."""
    rng = np.random.default_rng(seed)
    src = [rng.integers(0, n, n * 3)]
    dst = [rng.integers(0, n, n * 3)]
    spokes = rng.choice(n, n // 2, replace=False)
    src += [np.zeros_like(spokes), spokes, np.arange(n)]
    dst += [spokes, np.zeros_like(spokes), (np.arange(n) + 1) % n]
    src, dst = np.concatenate(src), np.concatenate(dst)
    order = np.lexsort((dst, src))
    src, dst = src[order], dst[order]
    row_ptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(src, minlength=n), out=row_ptr[1:])
    return row_ptr, dst.astype(np.int64)
