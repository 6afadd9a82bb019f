#!/usr/bin/env python3
"""What the root of an N-GPU run does beside its own walk, rehearsed on one card: while the node2vec walk of one step
runs on one stream, the blocks of the previous step are widened (uint32 wire format -> int64 rows) on another.  The
widening kernels use no LDS, so they should find room on CUs the persistent walk kernel has filled -- unlike RCCL's
send/recv kernels, for which bench.py leaves 64 workgroup slots free (scripts/overlap_probe.py).  Prints the walk's and
the widening's durations alone and side by side, with and without the spare slots.
usage: python scripts/overlap_wire.py [peers]      (GPU box)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from torch_random_walk_amd import graphs, rw  # noqa: E402

peers = int(sys.argv[1]) if len(sys.argv) > 1 else 7
dev = torch.device("cuda", 0)
n, L = 10_000_000, 80
rp, ci = graphs.rmat(n, 100_000_000, seed=1, device=dev)
tg = torch.arange(n, device=dev)
rw.prepare(rp, ci, marks="full")
out = torch.empty((n, L + 1), dtype=torch.int64, device=dev)
land = torch.randint(0, n, (peers * n, L + 1), dtype=torch.int32, device=dev)     # what the peers sent
full = torch.empty((peers * n, L + 1), dtype=torch.int64, device=dev)
prio = int(os.environ.get("WIRE_PRIORITY", "0"))     # -1: the widening on a high-priority stream
s_walk, s_wire = torch.cuda.Stream(), torch.cuda.Stream(priority=prio)


def walk():
    rw.walk_ex(rp, ci, tg, 0.25, 4.0, L, 10, rng_mode=rw.RNG_WALKER, check=False, out=out)


def widen():
    for r in range(peers):
        rw.widen_rows(land[r * n:(r + 1) * n], out=full[r * n:(r + 1) * n])


def timed(fns):
    """fns: list of (stream, fn); all started together; returns each one's duration and the whole's"""
    torch.cuda.synchronize()
    evs = []
    start = torch.cuda.Event(enable_timing=True)
    start.record()
    for s, fn in fns:
        s.wait_event(start)
        with torch.cuda.stream(s):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(s)
            fn()
            b.record(s)
            evs.append((a, b))
    torch.cuda.synchronize()
    return [a.elapsed_time(b) for a, b in evs], max(start.elapsed_time(b) for _, b in evs)


for spare in ("0", "64"):
    os.environ["TRW_POOL_SPARE_BLOCKS"] = spare
    for _ in range(2):
        walk()
        widen()
    (w_alone,), _ = timed([(s_walk, walk)])
    (x_alone,), _ = timed([(s_wire, widen)])
    (w_both, x_both), whole = timed([(s_walk, walk), (s_wire, widen)])
    (x_both2, w_both2), whole2 = timed([(s_wire, widen), (s_walk, walk)])
    gb = peers * n * (L + 1) * 12 / 1e9
    print("wire stream priority %d, spare slots %2s: walk alone %.1f ms, widening %d blocks (%.0f GB moved) alone %.1f ms; together, walk queued first: "
          "walk %.1f widen %.1f whole %.1f ms; widening queued first: walk %.1f widen %.1f whole %.1f ms" % (
              prio, spare, w_alone, peers, gb, x_alone, w_both, x_both, whole, w_both2, x_both2, whole2), flush=True)

# ---- round 5: narrow rows end to end.  The root's result is int32; its own walk writes narrow rows into its slice and the
# peers' blocks LAND in theirs -- nothing is widened.  What shares the memory system with the walk is then only the
# ingest: 7 x 3.24 GB written per step (rehearsed as a device copy on another stream: it READS as much as it writes, so it is
# heavier than the links' DMA writes).
del full
torch.cuda.empty_cache()
res32 = torch.empty(((peers + 1) * n, L + 1), dtype=torch.int32, device=dev)


def walk32():
    rw.walk_ex(rp, ci, tg, 0.25, 4.0, L, 10, rng_mode=rw.RNG_WALKER, check=False, out=res32[:n])


def ingest():
    res32[n:].copy_(land)


os.environ["TRW_POOL_SPARE_BLOCKS"] = "64"
for _ in range(2):
    walk32()
    ingest()
(w64,), _ = timed([(s_walk, walk)])
(w32,), _ = timed([(s_walk, walk32)])
(i_alone,), _ = timed([(s_wire, ingest)])
(w32_both, i_both), whole = timed([(s_walk, walk32), (s_wire, ingest)])
assert torch.equal(rw.widen_rows(res32[:n].contiguous()), out), "narrow rows differ from the int64 walk"
print("narrow rows end to end (spare slots 64): walk writing int64 rows alone %.1f ms, writing uint32 rows alone %.1f ms; ingest of %d peer "
      "blocks (%.1f GB copied in) alone %.1f ms; together: walk %.1f ingest %.1f whole %.1f ms -- nothing to widen" % (
          w64, w32, peers, peers * n * (L + 1) * 4 / 1e9, i_alone, w32_both, i_both, whole), flush=True)
