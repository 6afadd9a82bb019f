import os, sys, ctypes
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from tests.test_gpu_prepared import _skewed_graph, _prepared_ext32, _dev
from torch_random_walk_amd import _native
L = _native.cabi()
n = 6000
rp, ci = _skewed_graph(n, 400_000, 5, hubs=((3, 17000), (11, 20000), (40, 5000), (41, 1100), (77, 300)))
nnz = len(ci)
d_rp, d_ci = _dev(rp, ci)
what = 6
os.environ["TRW_HEAVY_TILE"] = "1"
tiled = _prepared_ext32(L, d_rp, d_ci, n, nnz, what)
os.environ["TRW_HEAVY_TILE"] = "0"
queued = _prepared_ext32(L, d_rp, d_ci, n, nnz, what)
bad = np.nonzero((tiled != queued).any(1))[0]
deg = np.diff(rp); src = np.repeat(np.arange(n), deg)
print("mismatches", bad.size, "of", nnz)
def want(e):
    u, v = src[e], ci[e]
    row_u, row_v = ci[rp[u]:rp[u + 1]], ci[rp[v]:rp[v + 1]]
    bits = np.zeros(128, dtype=bool)
    np.logical_or.at(bits, np.nonzero(np.isin(row_v, row_u))[0] % 128, True)
    return np.packbits(bits, bitorder="little").view(np.uint32)
for e in bad[:25]:
    u, v = src[e], ci[e]
    print("e %d (e-rp[u] %d) u %d v %d deg_u %d deg_v %d\n   tiled  %s\n   queued %s\n   numpy  %s" % (e, e - rp[u], u, v, deg[u], deg[v],
          [hex(x) for x in tiled[e]], [hex(x) for x in queued[e]], [hex(x) for x in want(e)]))
import collections
print("by (deg_u>deg_v):", collections.Counter((deg[src[bad]] > deg[ci[bad]]).tolist()))
print("by (e - rp[u]) % 8:", collections.Counter(((bad - rp[src[bad]]) % 8).tolist()))
print("u of bad:", collections.Counter(src[bad].tolist()).most_common(8))
