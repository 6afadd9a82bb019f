"""GPU box: the prepare pass on Erdos-Renyi graphs -- nearly every row between the inline cap and 32 entries, or every row
heavy -- : how full the in-list and task rooms get, tile pass against the wave-per-entry pass (whose queue has room for fewer
entries), and the walk against the oracle.  usage: python scripts/check_prepare_er.py"""
import os, sys, ctypes
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from tests.test_gpu_prepared import _prepared_ext32, _dev
from torch_random_walk_amd import _native, graphs
from torch_rw import rw
from oracle import oracle as O
L = _native.cabi()
for name, (n, m) in (("er100k", (100_000, 1_000_000)), ("er30k_dense", (30_000, 1_500_000))):
    rp_t, ci_t = graphs.erdos_renyi(n, m, seed=3)
    rp, ci = rp_t.numpy(), ci_t.numpy()
    nnz = len(ci)
    d_rp, d_ci = _dev(rp, ci)
    deg = np.diff(rp)
    print(name, "nnz", nnz, "deg min/mean/max", deg.min(), deg.mean(), deg.max())
    os.environ["TRW_HEAVY_TILE"] = "1"
    tiled = _prepared_ext32(L, d_rp, d_ci, n, nnz, 14)
    pb = L.trw_prepare_csr_bytes(n, nnz, 14)
    blob = torch.empty(pb, dtype=torch.uint8, device="cuda:0")
    assert L.trw_prepare_csr_hip(d_rp.data_ptr(), n, d_ci.data_ptr(), nnz, 14, blob.data_ptr(), pb, torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    hdr = blob[:256].view(torch.int64).cpu().tolist()
    room_items = nnz // 5 * 3 + 1024
    print("  in-list items", hdr[17], "of room", room_items, "; tasks", hdr[16], "of room", nnz // 5 + 4096)
    os.environ["TRW_HEAVY_TILE"] = "0"
    queued = _prepared_ext32(L, d_rp, d_ci, n, nnz, 14)
    os.environ["TRW_HEAVY_TILE"] = "1"
    bad = np.nonzero((tiled != queued).any(1))[0]
    print("  entries where tile pass and wave-per-entry pass differ:", bad.size)
    allones = ((tiled[:, 4:] == 0xFFFFFFFF).all(1)).sum()
    print("  entries with all-ones masks (tiled):", int(allones), "(queued):", int(((queued[:, 4:] == 0xFFFFFFFF).all(1)).sum()))
    tg = np.random.default_rng(1).integers(0, n, 3000)
    ref = O.walk(rp, ci, tg, 0.25, 4.0, 20, 7, rng_mode=O.RNG_WALKER)
    rw.clear_cache()
    rw.prepare(d_rp, d_ci, marks="full")
    got = rw.walk_ex(d_rp, d_ci, torch.as_tensor(tg).cuda(), 0.25, 4.0, 20, 7, rng_mode=rw.RNG_WALKER).cpu().numpy()
    print("  walk equals oracle:", bool((got == ref).all()))
