#!/usr/bin/env python3
"""GPU box: prepares one graph several times (C ABI, fresh buffers poisoned differently each time) and demands the SAME
32-byte entries every time -- the marking passes take tasks and in-list places in whatever order the hardware hands them out,
the entries must not notice.  usage: scripts/check_prepare_repeat.py [nodes edges [reps [what]]]"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from torch_random_walk_amd import graphs, _native
n, m = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1_000_000, 10_000_000)
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
what = int(sys.argv[4]) if len(sys.argv) > 4 else 14
L = _native.cabi()
if os.environ.get('TRW_LIB'):
    # (a variant build: same prototypes)
    V = ctypes.CDLL(os.environ['TRW_LIB'])
    for name in ('trw_prepare_csr_bytes', 'trw_prepare_csr_region', 'trw_prepare_csr_hip'):
        getattr(V, name).argtypes = getattr(L, name).argtypes
        getattr(V, name).restype = getattr(L, name).restype
    L = V
dev = "cuda:0"
rp, ci = graphs.rmat(n, m, seed=1, device=dev)
nnz = ci.numel()
stream = torch.cuda.current_stream().cuda_stream
pb = L.trw_prepare_csr_bytes(n, nnz, what)
off, size = ctypes.c_size_t(), ctypes.c_size_t()
assert L.trw_prepare_csr_region(n, nnz, what, 5, ctypes.byref(off), ctypes.byref(size)) == 0
ref = None
for r in range(reps):
    blob = torch.full((pb,), (0x11 * (r + 1)) & 0xFF, dtype=torch.uint8, device=dev)
    assert L.trw_prepare_csr_hip(rp.data_ptr(), n, ci.data_ptr(), nnz, what, blob.data_ptr(), pb, stream) == 0
    torch.cuda.synchronize()
    ext = blob[off.value:off.value + size.value].view(torch.int32).view(nnz, 8).clone()
    hdr = blob[:256].view(torch.int64).tolist()
    if ref is None:
        ref = ext
        print("items %d dropped %d tasks %d + %d" % (hdr[20], hdr[19], hdr[16], hdr[21]), flush=True)
    else:
        bad = (ext != ref).any(1).nonzero().flatten()
        print("rep %d: %d entries differ %s" % (r, bad.numel(), bad[:6].tolist()), flush=True)
        src = None
        for e in bad[:6].tolist():
            if src is None:
                deg = rp[1:] - rp[:-1]
            u = int(torch.searchsorted(rp, torch.tensor([e], device=dev), right=True)[0]) - 1
            v = int(ci[e])
            row_u, row_v = ci[rp[u]:rp[u + 1]], ci[rp[v]:rp[v + 1]]
            hit = torch.isin(row_v, row_u).nonzero().flatten() % 128
            want = [0, 0, 0, 0]
            for b in hit.tolist():
                want[b >> 5] |= 1 << (b & 31)
            want = [w - (1 << 32) if w >= (1 << 31) else w for w in want]
            a, b, w = ext[e, 4:].tolist(), ref[e, 4:].tolist(), want
            kind = "THIS run wrong" if b == w else ("FIRST run wrong" if a == w else "both wrong")
            xor = [(x ^ y) & 0xFFFFFFFF for x, y in zip(a, b)]
            print("   entry %d: u %d (deg %d) -> v %d (deg %d); %s; differing bits %s; words 0..3 this %s first %s" % (
                e, u, int(deg[u]), v, int(deg[v]), kind, [hex(x) for x in xor], ext[e, :4].tolist(), ref[e, :4].tolist()))
    del blob
