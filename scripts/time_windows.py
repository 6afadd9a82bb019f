#!/usr/bin/env python3
"""Device time of the four window builders on BASELINE-scale inputs (row f-3); run on the GPU box.
usage: python scripts/time_windows.py [num_walks]"""
import sys
import os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_random_walk_amd import rw, graphs  # noqa: E402

dev = "cuda:0"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        out = fn()
    b.record()
    torch.cuda.synchronize()
    nbytes = sum(t.numel() * 8 for t in out)
    return a.elapsed_time(b) / reps, nbytes


g = torch.Generator(device=dev).manual_seed(1)
walks = torch.randint(0, 10_000_000, (N, 81), generator=g, device=dev)
ms, nb = timed(lambda: rw.to_windows(walks, 5, 10_000_000, 10))
print("to_windows              %d walks x 81, ws=5: %.3f ms, %.2f GB out + %.2f GB in -> %.0f GB/s" % (N, ms, nb / 1e9, walks.numel() * 8 / 1e9, (nb + walks.numel() * 8) / ms / 1e6))
rw.set_rng_mode(rw.RNG_WALKER)
ms, nb = timed(lambda: rw.to_windows_cbow(walks, 5, 10_000_000, 10))
print("to_windows_cbow         %d walks x 81, ws=5: %.3f ms, %.2f GB out -> %.0f GB/s" % (N, ms, nb / 1e9, (nb + walks.numel() * 8) / ms / 1e6))
rw.set_rng_mode(rw.RNG_AUTO)
triples, ents, pad = graphs.synthetic_kg(1_000_000, 5_000_000, 1000, seed=1, device=dev)
tw = torch.randint(0, 1_000_000, (N // 4, 161), generator=g, device=dev)
ms, nb = timed(lambda: rw.to_windows_triples(tw, 2, 1_001_000, pad, triples, 10))
print("to_windows_triples      %d walks x 161, ws=2: %.3f ms, %.2f GB out -> %.0f GB/s" % (N // 4, ms, nb / 1e9, (nb + tw.numel() * 8) / ms / 1e6))
ms, nb = timed(lambda: rw.to_windows_triples_ex(tw, 2, pad, triples, 10, cbow=True, base=0, rng_mode=rw.RNG_WALKER))
print("to_windows_triples_cbow %d walks x 161, ws=2: %.3f ms, %.2f GB out -> %.0f GB/s" % (N // 4, ms, nb / 1e9, (nb + tw.numel() * 8) / ms / 1e6))

# ---- the fused walk + to_windows against the two calls (R-MAT 1 M / 10 M, one walker per node, L = 80, window 5) ----
n = 1_000_000
rp, ci = graphs.rmat(n, 10_000_000, seed=1, device=dev)
tg = torch.arange(n, device=dev)
for _ in range(3):          # (the cache builds the 16-byte entries over the first calls)
    rw.walk(rp, ci, tg, 1.0, 1.0, 80, 10)
ms_w, _ = timed(lambda: (rw.walk(rp, ci, tg, 1.0, 1.0, 80, 10),))
wk = rw.walk(rp, ci, tg, 1.0, 1.0, 80, 10)
ms_t, nb = timed(lambda: rw.to_windows(wk, 5, n, 10))
ms_2, _ = timed(lambda: rw.to_windows(rw.walk(rp, ci, tg, 1.0, 1.0, 80, 10), 5, n, 10))
ms_f, _ = timed(lambda: rw.walk_windows(rp, ci, tg, 1.0, 1.0, 80, 10, 5, n))
ms_fw, _ = timed(lambda: rw.walk_windows(rp, ci, tg, 1.0, 1.0, 80, 10, 5, n, return_walks=True))
print("walk_windows  1M walkers L=80 ws=5: walk %.3f ms + to_windows %.3f ms = %.3f (two calls back to back %.3f); "
      "fused %.3f ms (%.2fx of the two calls), fused + walks kept %.3f ms; %.2f GB of windows" % (
          ms_w, ms_t, ms_w + ms_t, ms_2, ms_f, ms_f / ms_2, ms_fw, nb / 1e9))
