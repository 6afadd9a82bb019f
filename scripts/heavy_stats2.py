"""Workload of the neighbourhood-mask pass, split the way a row-in-LDS design would see it:
for every entry u -> v: shorter / longer row, which of the two is the longer one (u: entries are consecutive in
the CSR; v: needs the in-lists), and how long the longer row is (does it fit an LDS tile)."""
import sys, time
import torch
sys.path.insert(0, '/root/repo')
from torch_random_walk_amd import graphs
n = 10_000_000
t = time.time()
rp, ci = graphs.rmat(n, 100_000_000, seed=1)
print("gen %.0f s, nnz %d" % (time.time() - t, ci.numel()))
deg = rp[1:] - rp[:-1]
src = torch.repeat_interleave(torch.arange(n), deg)
du = deg[src]; dv = deg[ci]
sh = torch.minimum(du, dv); lg = torch.maximum(du, dv)
u_longer = du > dv
print("max degree %d; nodes with deg > 32: %d, > 1024: %d, > 8192: %d, > 16384: %d, > 65536: %d" % (
    int(deg.max()), int((deg > 32).sum()), int((deg > 1024).sum()), int((deg > 8192).sum()), int((deg > 16384).sum()), int((deg > 65536).sum())))
for lo, hi in [(0, 32), (32, 256), (256, 1024)]:
    for name, side in (("u longer (by CSR row)", u_longer), ("v longer/equal (by in-list)", ~u_longer)):
        m0 = (sh > lo) & (sh <= hi) & side
        print("shorter in (%d,%d], %s: items %.3e  sum shorter %.3e" % (lo, hi, name, int(m0.sum()), float(sh[m0].sum())))
        for llo, lhi in [(0, 2048), (2048, 8192), (8192, 16384), (16384, 65536), (65536, 1 << 40)]:
            m = m0 & (lg > llo) & (lg <= lhi)
            k = int(m.sum())
            if k:
                print("      longer in (%6d,%8d]: items %.3e  sum shorter %.3e  sum longer %.3e" % (
                    llo, min(lhi, int(deg.max())), k, float(sh[m].sum()), float(lg[m].sum())))
# per-row view for the in-list design: how many heavy in-list items does a node get
m = (sh > 32) & (sh <= 1024) & ~u_longer
cnt = torch.bincount(ci[m], minlength=n)
print("in-lists (v longer, shorter in (32,1024]): nodes with items %d, max items %d, items on nodes with > 4096 items: %.3e of %.3e" % (
    int((cnt > 0).sum()), int(cnt.max()), float(cnt[cnt > 4096].sum()), float(cnt.sum())))
