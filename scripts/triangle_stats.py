#!/usr/bin/env python3
"""For the (t, prev) pairs a biased walk actually visits: how often do row(t) and row(prev) share no node?
(Then every band trial of that step -- draw in [lo, hi): accept iff candidate in row(t) -- rejects without
looking at anything.)  Also: how many probes finding out costs per edge.  Run on the GPU box."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_random_walk_amd import rw, graphs  # noqa: E402

dev = "cuda:0"
n, m = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (10_000_000, 100_000_000)
rp, ci = graphs.rmat(n, m, seed=1, device=dev)
tg = torch.arange(n, device=dev)[:: max(1, n // 200_000)]
w = rw.walk_ex(rp, ci, tg, 0.25, 4.0, 80, 10, rng_mode=rw.RNG_WALKER).cpu().numpy()
rp_h, ci_h = rp.cpu().numpy(), ci.cpu().numpy()
rng = np.random.default_rng(0)
rows = rng.integers(0, w.shape[0], 20000)
cols = rng.integers(0, 79, 20000)
empty = 0
cost = 0
deg_t, deg_p, is_empty = [], [], []
for r, c in zip(rows, cols):
    t, p = w[r, c], w[r, c + 1]
    a = ci_h[rp_h[t]:rp_h[t + 1]]
    b = ci_h[rp_h[p]:rp_h[p + 1]]
    deg_t.append(len(a))
    deg_p.append(len(b))
    small, big = (a, b) if len(a) <= len(b) else (b, a)
    hit = np.isin(small, big)
    is_empty.append(not hit.any())
    if hit.any():
        cost += int(np.argmax(hit)) + 1
    else:
        empty += 1
        cost += len(small)
mind = np.minimum(np.array(deg_t), np.array(deg_p))
is_empty = np.array(is_empty)
for K in (1, 2, 4, 8, 16, 32, 64):
    sel = mind <= K
    print("visited pairs with min degree <= %d: %.1f %% of pairs, of which empty %.1f %% (-> %.1f %% of all pairs decided cheaply)"
          % (K, 100.0 * sel.mean(), 100.0 * is_empty[sel].mean() if sel.any() else 0.0, 100.0 * (sel & is_empty).mean()))
print("visited (t, prev) pairs with row(t) & row(prev) empty: %.1f %%; probes to find out (early exit, smaller row first): "
      "%.1f per pair; median deg t %d prev %d, mean %.0f / %.0f"
      % (100.0 * empty / len(rows), cost / len(rows), np.median(deg_t), np.median(deg_p), np.mean(deg_t), np.mean(deg_p)))
# the same over ALL edges (what a prepare pass would pay): sample of edges
e = rng.integers(0, len(ci_h), 20000)
src = np.searchsorted(rp_h, e, side="right") - 1
empty_e = 0
cost_e = 0
mind_e = []
for u, v in zip(src, ci_h[e]):
    a = ci_h[rp_h[u]:rp_h[u + 1]]
    b = ci_h[rp_h[v]:rp_h[v + 1]]
    small, big = (a, b) if len(a) <= len(b) else (b, a)
    mind_e.append(len(small))
    hit = np.isin(small, big)
    if hit.any():
        cost_e += int(np.argmax(hit)) + 1
    else:
        empty_e += 1
        cost_e += len(small)
print("all edges: empty %.1f %%, probes %.1f per edge" % (100.0 * empty_e / len(e), cost_e / len(e)))
mind_e = np.array(mind_e)
for K in (1, 2, 4, 8, 16, 32, 64):
    sel = mind_e <= K
    print("all edges with min degree <= %d: %.1f %%, mean smaller row %.2f entries (probes per such edge, upper bound)"
          % (K, 100.0 * sel.mean(), mind_e[sel].mean() if sel.any() else 0.0))
