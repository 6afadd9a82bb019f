import os, sys, ctypes
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from tests.test_gpu_prepared import _skewed_graph, _prepared_ext32, _dev
from torch_random_walk_amd import _native
L = _native.cabi()
n = 40000
rp, ci = _skewed_graph(n, 500_000, 5, hubs=((3, 17000), (11, 20000), (40, 5000), (41, 1100), (77, 300)))
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
# per tile node: how many of its heavy entries are bad
cap = 256
sh = np.minimum(deg[src], deg[ci]); heavy = (sh > 32) & (sh <= cap)
tile_node = np.where(deg[src] > deg[ci], src, ci)
badmask = np.zeros(nnz, bool); badmask[bad] = True
tn_bad = collections.Counter(tile_node[bad].tolist())
print("tile nodes with bad entries:", len(tn_bad))
rows = []
for tn, k in sorted(tn_bad.items())[:40]:
    tot = int((heavy & (tile_node == tn)).sum())
    a_tot = int((heavy & (tile_node == tn) & (deg[src] > deg[ci])).sum()); a_bad = int((badmask & (tile_node == tn) & (deg[src] > deg[ci])).sum())
    rows.append((tn, int(deg[tn]), k, tot, a_bad, a_tot))
print("(tile node, deg, bad, heavy entries, bad A, heavy A):", rows)
print("tile node ids min/max", min(tn_bad), max(tn_bad), " tile nodes // 64:", sorted(collections.Counter([t // 64 for t in tn_bad]).items())[:40])
# the task records of the tiled prepare
os.environ["TRW_HEAVY_TILE"] = "1"
pb = L.trw_prepare_csr_bytes(n, nnz, what)
blob = torch.empty(pb, dtype=torch.uint8, device="cuda:0")
assert L.trw_prepare_csr_hip(d_rp.data_ptr(), n, d_ci.data_ptr(), nnz, what, blob.data_ptr(), pb, torch.cuda.current_stream().cuda_stream) == 0
torch.cuda.synchronize()
hdr = blob[:256].view(torch.int64).cpu().tolist()
print("header slots 11,16..20:", hdr[11], hdr[16:21])
off, size = ctypes.c_size_t(), ctypes.c_size_t()
assert L.trw_prepare_csr_region(n, nnz, what, 6, ctypes.byref(off), ctypes.byref(size)) == 0
tasks = blob[off.value:off.value + size.value].cpu().numpy().view(np.uint32).reshape(-1, 4)
nt = hdr[19]
tk = tasks[:nt]
print("tasks:", nt, "room", tasks.shape[0], "first", tk[:6].tolist())
have = set(tk[:, 0].tolist())
missing = [t for t in sorted(tn_bad) if t not in have]
print("bad tile nodes without any task:", len(missing), missing[:20])
cnt_nodes = collections.Counter(tk[:, 0].tolist())
print("nodes with tasks:", len(cnt_nodes), "records beyond count that look written:", int((tasks[nt:nt + 100000, 2] != 0).sum()))
w12 = tk[(tk[:, 0] >= 768) & (tk[:, 0] < 832)]
print("tasks of nodes 768..831:", w12[:20].tolist())
