import torch, time, sys
sys.path.insert(0,'/root/repo')
from torch_random_walk_amd import graphs
t=time.time()
rp,ci=graphs.rmat(10_000_000,100_000_000,seed=1)
print("gen",time.time()-t, ci.numel())
deg=(rp[1:]-rp[:-1])
src=torch.repeat_interleave(torch.arange(10_000_000),deg)
du=deg[src]; dv=deg[ci]
sh=torch.minimum(du,dv); lg=torch.maximum(du,dv)
for lo,hi in [(0,32),(32,256),(256,1024),(1024,1<<40)]:
    m=(sh>lo)&(sh<=hi)
    n=int(m.sum())
    print("shorter in (%d,%d]: items %.3e  sum shorter %.3e  mean log2(longer/64) %.2f  sum shorter*log2 %.3e"%(lo,hi,n,float(sh[m].sum()),float(torch.log2(lg[m].float()/64).clamp(min=0).mean()) if n else 0, float((sh[m].float()*torch.log2(lg[m].float()/64).clamp(min=1)).sum())))
