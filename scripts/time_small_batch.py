#!/usr/bin/env python3
"""AUTO-mode small batches: a biased walk of N <= 32768/(L+1) walkers replays the
reference's serial glibc chain.  Times 404 and 405 walkers on R-MAT 10M/110M, cold and from the cache."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from torch_random_walk_amd import graphs, rw  # noqa: E402

dev = torch.device("cuda", 0)
rp, ci = graphs.rmat(10_000_000, 100_000_000, seed=1, device=dev)
tg = torch.randint(0, 10_000_000, (405,), device=dev, generator=torch.Generator(device=dev).manual_seed(3))
# the same replay over a graph prepared in full (rw.prepare(marks="full")): fewer candidates the masks cannot decide
rw.clear_cache()
rw.prepare(rp, ci, marks="full")
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    w = rw.walk(rp, ci, tg[:404], 0.25, 4.0, 80, 10)
    torch.cuda.synchronize()
    print("%-22s N=%d call %d: %.2f ms  cache %s" % ("replay, marks=full", 404, rep + 1, (time.perf_counter() - t0) * 1e3,
                                               [(i[2], i[4]) for i in rw.cache_info()]), flush=True)
for env in ({}, {"TRW_WAVE_REPLAY": "0"}):
    os.environ.pop("TRW_WAVE_REPLAY", None)
    os.environ.update(env)
    rw.clear_cache()
    for n in (404, 405):
        for rep in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            w = rw.walk(rp, ci, tg[:n], 0.25, 4.0, 80, 10)
            torch.cuda.synchronize()
            print("%-22s N=%d call %d: %.2f ms  cache %s" % (env or "wave replay", n, rep + 1, (time.perf_counter() - t0) * 1e3,
                                                       [(i[2], i[4]) for i in rw.cache_info()]), flush=True)
