#!/usr/bin/env python3
"""Does a small-grid kernel on a second stream (what an RCCL send/recv group is) run WHILE the persistent
biased walk kernel holds every wave slot of the chip?  One GPU; the copy stands in for the gather of step k,
the walk is step k+1, both become runnable when step k finishes -- exactly bench.py's pipeline at N > 1.
usage (GPU box): hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o /tmp/liboverlap_probe.so scripts/overlap_probe.hip
                 python scripts/overlap_probe.py [blocks] [copy_GB]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_random_walk_amd import rw, graphs  # noqa: E402

blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 32
gb = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
lds = int(sys.argv[3]) if len(sys.argv) > 3 else 0        # LDS bytes each side-stream workgroup claims
lib = ctypes.CDLL("/tmp/liboverlap_probe.so")
lib.probe_copy_lds.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]


def _copy(src_p, dst_p, nbytes, nblocks, stream_p):
    return lib.probe_copy_lds(src_p, dst_p, nbytes, nblocks, lds, stream_p)


lib_probe_copy = _copy
dev = "cuda:0"
rp, ci = graphs.rmat(10_000_000, 100_000_000, seed=1, device=dev)
tg = torch.arange(10_000_000, device=dev)
n = int(gb * (1 << 30)) // 8
src = torch.ones(n, dtype=torch.int64, device=dev)
dst = torch.zeros(n, dtype=torch.int64, device=dev)
side = torch.cuda.Stream()


rw.prepare(rp, ci, marks="full")


def walk():
    return rw.walk_ex(rp, ci, tg, 0.25, 4.0, 80, 10, rng_mode=rw.RNG_WALKER, check=False)


def ev():
    return torch.cuda.Event(enable_timing=True)


walk()
torch.cuda.synchronize()
# standalone durations
a, b = ev(), ev()
a.record()
walk()
b.record()
torch.cuda.synchronize()
t_walk = a.elapsed_time(b)
with torch.cuda.stream(side):
    a, b = ev(), ev()
    a.record()
    lib_probe_copy(src.data_ptr(), dst.data_ptr(), n * 8, blocks, side.cuda_stream)
    b.record()
torch.cuda.synchronize()
t_copy = a.elapsed_time(b)
# the pipeline: walk k; then {copy on the side stream, walk k+1} both waiting for walk k
w0, w1, w2, c0, c1, done_k = ev(), ev(), ev(), ev(), ev(), torch.cuda.Event()
w0.record()
walk()
done_k.record()
w1.record()
with torch.cuda.stream(side):
    side.wait_event(done_k)
    c0.record()
    lib_probe_copy(src.data_ptr(), dst.data_ptr(), n * 8, blocks, side.cuda_stream)
    c1.record()
walk()
w2.record()
torch.cuda.synchronize()
print("OVERLAP blocks=%d lds=%d B copy=%.1f GB: walk alone %.1f ms, copy alone %.1f ms (%.0f GB/s); pipelined: walk k+1 %.1f ms, "
      "copy %.1f ms, copy finished %.1f ms after walk k+1 started (walk k+1 took %.1f)"
      % (blocks, lds, gb, t_walk, t_copy, 2 * n * 8 / t_copy / 1e6, w1.elapsed_time(w2), c0.elapsed_time(c1),
         w1.elapsed_time(c1), w1.elapsed_time(w2)))
