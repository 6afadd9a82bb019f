"""Degree statistics of the nodes a walk actually stands on (GPU box helper)."""
import sys, torch
sys.path.insert(0, ".")
from torch_random_walk_amd import graphs
from torch_rw import rw
dev = "cuda:0"
rp, ci = graphs.rmat(10_000_000, 100_000_000, seed=1, device=dev)
deg = rp[1:] - rp[:-1]
print("nodes", deg.numel(), "nnz", ci.numel(), "max deg", int(deg.max()), "median", int(deg.median()), "mean", float(deg.float().mean()))
for cap in (4, 8, 16, 32, 64, 128, 1024):
    print("static: frac nodes deg<=%d: %.4f   frac entries in such rows: %.4f" % (cap, float((deg <= cap).float().mean()), float(deg[deg <= cap].sum()) / ci.numel()))
tg = torch.arange(2_000_000, device=dev)
for p, q in ((1.0, 1.0), (0.25, 4.0)):
    w = rw.walk_ex(rp, ci, tg, p, q, 80, 10, rng_mode=(1 if p == 1.0 else 2))
    d = deg[w[:, 1:-1].reshape(-1)]
    print("p=%g q=%g visited-node out-degree: mean %.1f median %d" % (p, q, float(d.float().mean()), int(d.median())))
    for cap in (4, 8, 16, 32, 64, 128, 1024):
        print("   frac visited with deg<=%d: %.4f" % (cap, float((d <= cap).float().mean())))
