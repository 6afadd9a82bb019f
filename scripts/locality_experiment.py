#!/usr/bin/env python3
"""Locality experiment for the gather-bound walks (VERDICT r01 "what's weak" 5): what would re-ordering walkers by
the node they stand on buy, and what would it cost?

A walk step gathers one fat entry of the row of the walker's current node.  After the first step the current nodes
of a batch are a random sample of the graph (weighted by in-degree), so consecutive lanes hit unrelated lines: the
L2 hit rate of the uniform walk is 7-8 %.  Sorting the batch by current node before EVERY step is the best case of
every bucketing scheme (options (a)/(b) of the verdict: finer order than any bucket, applied at every step).  This
script measures that best case with the production kernel itself:

  one-step walks (L=1) over 10 M walkers whose start nodes are (1) sorted, (2) drawn like the current nodes of a
  long walk (by in-degree) and shuffled, (3) the same sample sorted -- kernel time from the library's events;
  the price of the re-ordering: sorting 10 M (node, walker) pairs with torch.sort (rocPRIM radix sort), and
  gathering the 124-byte RNG windows / 4-byte ids through the permutation.

Run under rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum for the hit rates (scripts/locality_experiment.sh).
Prints a table; copy into profiles/.
"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from torch_random_walk_amd import _native, graphs, rw  # noqa: E402

dev = torch.device("cuda", 0)
lib = _native.cabi()
n, m, N = 10_000_000, 100_000_000, 10_000_000
rp, ci = graphs.rmat(n, m, seed=1, device=dev)
nnz = ci.numel()
g = torch.Generator(device=dev).manual_seed(5)
# current nodes of a long walk ~ a random edge's head... the node a walker stands on after a step is the TAIL of a
# uniformly chosen out-edge of its previous node; in the stationary regime that is close to sampling col_idx uniformly
cur = ci[torch.randint(0, nnz, (N,), generator=g, device=dev)]
cases = {
    "start nodes 0..N-1 (sorted, one per node)": torch.arange(N, device=dev),
    "in-degree-weighted sample, shuffled (= a step in mid-walk)": cur,
    "the same sample, sorted by node (= perfect re-ordering)": torch.sort(cur).values,
}
os.environ["TRW_EXT"] = "1"          # the fat-entry kernel, as the 10 M batch uses
rw.set_cache(True, 2)
rw.prepare(rp, ci, biased=False, marks=False, uniform=True)
lib.trw_profile_enable(1)
print("# uniform walk, ONE step per walker, %d walkers, R-MAT %d/%d, fat entries" % (N, n, nnz))
for L in (1, 2):
    for name, tg in cases.items():
        ts = []
        for _ in range(4):
            rw.walk_ex(rp, ci, tg, 1.0, 1.0, L, 10, rng_mode=rw.RNG_GLIBC, check=False)
            a, b, c = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
            lib.trw_profile_last(ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
            ts.append(b.value)
        print("L=%d  %-62s walk kernel %.3f ms  (%.2f ns per walker-step)" % (L, name, min(ts), min(ts) * 1e6 / (N * L)))
lib.trw_profile_enable(0)
# the price of re-ordering once
torch.cuda.synchronize()
for label, fn in (
        ("torch.sort of 10 M int64 node ids (+ permutation)", lambda: torch.sort(cur)),
        ("torch.sort of 10 M int32 node ids (+ permutation)", lambda: torch.sort(cur.to(torch.int32))),
        ("gather of 10 M x 124-byte RNG windows through a permutation", None),
        ("gather of 10 M x 4-byte walker ids through a permutation", None)):
    if fn is None:
        perm = torch.randperm(N, device=dev)
        src = torch.empty((N, 31 if "124" in label else 1), dtype=torch.int32, device=dev)
        fn = lambda src=src, perm=perm: src[perm]   # noqa: E731
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    print("cost  %-62s %.3f ms" % (label, (time.perf_counter() - t0) / 5 * 1e3))
