#!/usr/bin/env python3
"""bench.py -- walk-steps/s of the CSR random-walk hot path on 1..8 MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch: every rank walks one walk
of `walk_length` steps from each of its start nodes (weak scaling: the batch per
GPU is fixed; rank r's walkers carry global indices r*B .. r*B+B-1, i.e. the
job is "world_size walks per node", node2vec's `r` parameter), and for N > 1
the row blocks are gathered on rank 0 with ONE RCCL collective, inside the
timed region.  Inputs are resident in HBM before timing starts.

Workloads (BASELINE.json `configs`):
  rmat10m_biased   configs[2]  RMAT 10M nodes / 100M edges, node2vec p=0.25 q=4, L=80   (default:
                               the graph the north-star target is quoted on)
  rmat10m_uniform  same graph, p=q=1 (reference-exact glibc draws)
  er100k_uniform   configs[1]  Erdos-Renyi 100k nodes / 1M edges, p=q=1, L=80, 10 walks per node
  papers100m_uniform configs[3] R-MAT 111M nodes / 1.6B edges, p=q=1, L=80, 13.9M start nodes per GPU
  kg5m_triples     configs[4]  synthetic KG, 5M triples / 1M entities / 1000 relations, walk_triples, L=80

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the walk
kernel), its duration measured with HIP events on the launch stream through the
library's own instrumentation hook; `cpu_baseline` times the oracle's
libc-rand() single-thread mode (the reference's csrc/cpu algorithm) on a bounded
sample of the same workload.
"""
import argparse
import ctypes
import json
import os
import sys
import time

# multi-process GPU work on this pool needs dmabuf IPC (set before HIP starts; the driver exports it already)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md; a float4 device copy sustains 6.29 TB/s there)

WORKLOADS = {
    "rmat10m_biased": dict(graph="rmat", nodes=10_000_000, edges=100_000_000, p=0.25, q=4.0, walks_per_node=1),
    "rmat10m_uniform": dict(graph="rmat", nodes=10_000_000, edges=100_000_000, p=1.0, q=1.0, walks_per_node=1),
    "er100k_uniform": dict(graph="er", nodes=100_000, edges=1_000_000, p=1.0, q=1.0, walks_per_node=10),
    # BASELINE configs[3]: ogbn-papers100M-scale synthetic CSR, uniform; each GPU walks one eighth of the start
    # nodes (at 8 GPUs: the whole graph once), CSR replicated
    "papers100m_uniform": dict(graph="rmat", nodes=111_000_000, edges=1_600_000_000, p=1.0, q=1.0, walks_per_node=1,
                               shard_of=8),
    # the node2vec walk at that scale: one GPU's shard of the start nodes, p=0.25 q=4 -- the edge
    # table has 2^32 slots (64-bit slot arithmetic), prepared graph 131 GB beside the 14.6 GB CSR
    "papers100m_biased": dict(graph="rmat", nodes=111_000_000, edges=1_600_000_000, p=0.25, q=4.0, walks_per_node=1,
                              shard_of=8),
    # the same graph with ALL its start nodes on one GPU (72 GB of walks next to the 14 GB graph): what 288 GB allows
    "papers100m_full": dict(graph="rmat", nodes=111_000_000, edges=1_600_000_000, p=1.0, q=1.0, walks_per_node=1),
    # BASELINE configs[4]: knowledge-graph triples path (the reference's triples walk is uniform only,
    # torch_rw/rw.py:18 / rw_cpu_triples.cpp:105-127 -- there is no biased triples walk to reproduce)
    "kg5m_triples": dict(graph="kg", nodes=1_000_000, edges=5_000_000, relations=1000, p=1.0, q=1.0,
                         walks_per_node=1),
    # rows f-2 / f-3 of the scope table (edge-list walk, window builders)
    "elist_rmat1m_biased": dict(graph="elist", nodes=1_000_000, edges=10_000_000, p=0.25, q=4.0, walks_per_node=1),
    "elist_rmat1m_uniform": dict(graph="elist", nodes=1_000_000, edges=10_000_000, p=1.0, q=1.0, walks_per_node=1),
    "windows_1m": dict(graph="windows", nodes=10_000_000, edges=0, p=1.0, q=1.0, walks_per_node=1, walks=1_000_000,
                       window_size=5),
    # rw.walk_windows: the uniform walk and to_windows as one call (R-MAT 1 M / 10 M, one walker per node, window 5)
    "walk_windows_1m": dict(graph="rmat", nodes=1_000_000, edges=10_000_000, p=1.0, q=1.0, walks_per_node=1,
                            fused_window=5),
    # small variants for quick checks / CI on the GPU box
    "rmat1m_biased": dict(graph="rmat", nodes=1_000_000, edges=10_000_000, p=0.25, q=4.0, walks_per_node=1),
    "rmat1m_uniform": dict(graph="rmat", nodes=1_000_000, edges=10_000_000, p=1.0, q=1.0, walks_per_node=1),
}


def _self_launch(n):
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="rmat10m_biased", choices=sorted(WORKLOADS))
    ap.add_argument("--walk-length", type=int, default=80)
    ap.add_argument("--seed", type=int, default=10)
    ap.add_argument("--graph-seed", type=int, default=1)
    ap.add_argument("--no-gather", action="store_true", help="N>1: leave the row blocks on their ranks")
    ap.add_argument("--gather-chunks", type=int, default=4,
                    help="N>1, CSR walks: row chunks per step; each chunk's transfer starts when its walk is done")
    ap.add_argument("--wire", default="auto", choices=["auto", "int64"],
                    help="N>1: what the row blocks travel as (auto: uint32 while node ids fit, widened on arrival)")
    ap.add_argument("--result", default="int64", choices=["int64", "int32"],
                    help="N>1, CSR walks: dtype of the gathered [N, L+1] result on the root.  int32 (uint32 bit patterns: node "
                         "ids are < 2^32) = narrow rows end to end: every rank's walk kernel writes them, the peers' blocks land "
                         "in the result, nothing is widened.  The N = 1 line is always int64 (the reference's rows)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo is a dry-run aid for the N>1 control flow on a single-GPU box (all ranks share "
                         "cuda:0 and the gather is staged through host memory); its numbers mean nothing")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the baseline sample")
    ap.add_argument("--steady", action="store_true",
                    help="profiling runs: no cold first call and no checked re-run of a small batch, so that every launch "
                         "of the walk kernel in a rocprofv3 trace is a steady-state one (scripts/profile_gpu.sh)")
    ap.add_argument("--marks", default="full", choices=["full", "default", "none"],
                    help="CSR biased walk: how far the graph is prepared before the timed steps (rw.prepare marks=)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # Plain `python bench.py --gpus N`: start the N ranks ourselves, one process per GPU, as a child
        # `python -m torch.distributed.run` -- before this process has touched torch or the GPU (a process
        # that has initialised HIP must never exec or fork GPU work), and pass the child's exit code on.
        sys.exit(_self_launch(args.gpus))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.exit("bench.py --gpus %d does not match WORLD_SIZE=%d" % (args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    dry = args.backend == "gloo"
    if dry:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # the persistent walk kernel leaves 64 of its 1024 workgroup slots empty so that the gather's RCCL kernels get
        # on the chip WHILE the next step walks (profiles/r02_overlap_probe.txt: with every slot taken a side-stream
        # kernel starts only when the walk is over; with 64 spare it runs beside it, walk 48 -> 56 ms)
        os.environ.setdefault("TRW_POOL_SPARE_BLOCKS", "64")
        if dry:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from torch_random_walk_amd import _native, graphs, rw
    from torch_random_walk_amd.dist import shard_bounds  # noqa: F401  (the partition rule this bench follows)

    cfg = WORKLOADS[args.workload]
    L = args.walk_length
    p, q = cfg["p"], cfg["q"]
    uniform = (p == 1.0 and q == 1.0)
    n = cfg["nodes"]
    kg = cfg["graph"] == "kg"
    lib = _native.cabi()

    # ---- synthetic graph, generated on the device from a seed (identical on every rank) ----
    t0 = time.time()
    elist = cfg["graph"] == "elist"
    windows = cfg["graph"] == "windows"
    if kg:
        from torch_random_walk_amd import utils as trw_utils
        triples, entities, pad = graphs.synthetic_kg(n, cfg["edges"], cfg["relations"], seed=args.graph_seed, device=dev)
        rt_index, triples = trw_utils.build_relation_tail_index(triples, entities)
        nnz = int(triples.shape[0])
        cols = 2 * L + 1
    elif elist:
        # the same R-MAT edges as the CSR workloads, as an edge list sorted by head + (first,last) index;
        # no ring here: dead ends exercise the padding / restart path
        from torch_random_walk_amd import utils as trw_utils
        rp_tmp, ci_tmp = graphs.rmat(n, cfg["edges"], seed=args.graph_seed, device=dev)
        deg_tmp = rp_tmp[1:] - rp_tmp[:-1]
        heads = torch.repeat_interleave(torch.arange(n, device=dev), deg_tmp)
        keep = ci_tmp != (heads + 1) % n                      # drop the ring edges again
        el_edges, el_index = None, None
        el_index, el_edges = trw_utils.build_node_edge_index(torch.stack([heads[keep], ci_tmp[keep]], 1),
                                                             torch.arange(n, device=dev))
        pad = n
        nnz = int(el_edges.shape[0])
        cols = L + 1
        del rp_tmp, ci_tmp, deg_tmp, heads, keep
    elif windows:
        g_w = torch.Generator(device=dev).manual_seed(args.graph_seed)
        win_walks = torch.randint(0, n, (cfg["walks"], L + 1), generator=g_w, device=dev)
        nnz = 0
        cols = L + 1
    elif cfg["graph"] == "rmat":
        row_ptr, col_idx = graphs.rmat(n, cfg["edges"], seed=args.graph_seed, device=dev)
    else:
        row_ptr, col_idx = graphs.erdos_renyi(n, cfg["edges"], seed=args.graph_seed, device=dev)
    if not (kg or elist or windows):
        nnz = int(col_idx.numel())
        cols = L + 1
    if cfg.get("shard_of"):
        per = (n + cfg["shard_of"] - 1) // cfg["shard_of"]
        targets = (torch.arange(per, device=dev, dtype=torch.int64) + rank * per) % n
    else:
        targets = torch.arange(n, device=dev, dtype=torch.int64).repeat(cfg["walks_per_node"])
    if windows:
        targets = targets[:cfg["walks"]]
    B = int(targets.numel())          # walkers (walks, for the window builder) per GPU: fixed, weak scaling
    walker_base = rank * B            # global walker index of this rank's first walker
    torch.cuda.synchronize()
    t_graph = time.time() - t0
    if world > 1:
        # every rank generated its replica from the same seed: they must be the same graph
        if kg:
            chk = triples.sum() + rt_index.sum()
        elif elist:
            chk = el_edges.sum() + el_index.sum()
        elif windows:
            chk = win_walks.sum()
        else:
            chk = row_ptr.sum() * 31 + col_idx.sum()
        lo_hi = torch.stack([chk, -chk]).to(torch.int64)
        if dry:
            lo_hi = lo_hi.cpu()
        dist.all_reduce(lo_hi, op=dist.ReduceOp.MAX)
        assert int(lo_hi[0]) == -int(lo_hi[1]), "graph replicas differ across ranks"

    mode = rw.RNG_WALKER if (kg or elist or not uniform) else rw.RNG_GLIBC
    gather = (world > 1) and not args.no_gather
    full = None
    fulls, lands, wires, step_no = [], [], [], [0]
    csr = not (windows or elist or kg or cfg.get("fused_window"))
    if windows:
        cols = (L + 1 - cfg["window_size"] + 1) * (cfg["window_size"] - 1)
    if cfg.get("fused_window"):
        cols = (L + 1 - cfg["fused_window"] + 1) * (cfg["fused_window"] - 1)
    # What a block travels as (dist.wire_dtype_for): uint32 while every value fits -- node ids < n, the edge-list /
    # triples walks' padding id n, relation ids < n -- half the bytes per xGMI link; widened where it lands.
    wire32 = gather and args.wire != "int64" and (n + 1) <= (1 << 32)
    wire_bytes = 4 if wire32 else 8
    # The CSR walk is made in row chunks, each walked straight into its slice (walk_ex(out=...)) and handed to the
    # transport as soon as it is enqueued: the gather of a step overlaps the rest of the SAME step's walk, and the
    # next step's.  Two sets of buffers alternate because step k's blocks are still landing while step k+1 walks.
    chunks = max(1, args.gather_chunks) if (gather and csr) else 1
    # narrow rows end to end (--result int32): the root's result is int32, its own walk writes narrow rows into it and the
    # peers' blocks land in it -- no widening pass on the root (22.7 GB in, 45.4 GB out per step at N = 8, beside its walk)
    result32 = gather and csr and wire32 and args.result == "int32"
    # a peer's CSR walk writes its wire rows itself (walk_ex(out=<int32>)): no int64 block, no narrowing pass
    direct32 = gather and csr and wire32
    side = torch.cuda.Stream(device=dev) if gather else None
    if gather and rank == 0:
        fulls = [torch.empty((world * B, cols), dtype=torch.int32 if result32 else torch.int64, device=dev) for _ in range(2)]
        if (wire32 and not result32) or dry:
            lands = [torch.empty((world * B, cols), dtype=torch.int32 if wire32 else torch.int64,
                                 device="cpu" if dry else dev) for _ in range(2)]
        full = fulls[0]
    elif gather:
        if csr and not direct32:
            fulls = [torch.empty((B, cols), dtype=torch.int64, device=dev) for _ in range(2)]
        if wire32:
            wires = [torch.empty((B, cols), dtype=torch.int32, device=dev) for _ in range(2)]

    fused_ws = cfg.get("fused_window", 0)

    def walk(tg, base, check):
        if fused_ws:  # walk + to_windows in one call; the result the bench tracks is pos_windows, one row per walker
            return rw.walk_windows(row_ptr, col_idx, tg, p, q, L, args.seed, fused_ws, n, walker_base=base,
                                   check=check)[1].view(tg.numel(), -1)
        if windows:   # "walkers" are walks here; the result the bench tracks is the negatives tensor
            lo = int(tg[0]) if tg.numel() else 0
            return rw.to_windows_ex(win_walks[lo:lo + tg.numel()], cfg["window_size"], n, args.seed,
                                    walk_base=base + lo)[2].view(tg.numel(), -1)
        if elist:
            return rw.walk_edge_list_ex(el_edges, el_index, tg, p, q, L, args.seed, pad, restart=True,
                                        walker_base=base, rng_mode=mode, check=check)
        if kg:
            return rw.walk_triples_ex(triples, rt_index, tg, L, pad, args.seed, walker_base=base, rng_mode=mode,
                                      check=check)
        return rw.walk_ex(row_ptr, col_idx, tg, p, q, L, args.seed, walker_base=base, rng_mode=mode, check=check)

    class _Pending:
        """The transfers of one step.  wait(): they have landed (and, on the root, have been widened into the int64
        result) as far as the streams are concerned; nothing here blocks the host under RCCL."""
        def __init__(self):
            self.works, self.keep, self.pieces, self.k = [], [], [], 0
            self.ev_first = self.ev_walked = self.ev_done = None

        def wait(self):
            if rank != 0:
                for w in self.works:
                    w.wait()
                return
            land = lands[self.k % 2] if lands else None
            dst = fulls[self.k % 2]
            with torch.cuda.stream(side):
                for w in self.works:
                    w.wait()           # RCCL: the side stream waits; gloo (dry run): the host does
                for lo_, hi_ in self.pieces:
                    if land is None:
                        continue
                    got = land[lo_:hi_].to(dev, non_blocking=True) if dry else land[lo_:hi_]
                    if wire32 and not result32:
                        rw.widen_rows(got, out=dst[lo_:hi_])
                    else:
                        dst[lo_:hi_].copy_(got)
                self.ev_done = torch.cuda.Event(enable_timing=True)
                self.ev_done.record(side)

    walk_events = []   # (start, end) event pairs of the walk calls of the timed steps (N > 1)

    def step_once(timed=False):
        """One step: this rank's walks and, for N > 1, its share of the gather (posted, not waited for)."""
        nonlocal full
        if not gather:
            return walk(targets, walker_base, False), None
        from torch_random_walk_amd.dist import chunk_bounds
        k = step_no[0]
        step_no[0] += 1
        pend = _Pending()
        pend.k = k
        main = torch.cuda.current_stream()
        if rank == 0:
            full = fulls[k % 2]
            main.wait_stream(side)   # step k-2's blocks were widened out of / into these two buffers on the side stream
        mine = (wires[k % 2] if (direct32 and rank != 0) else fulls[k % 2][:B]) if (csr or rank == 0) else None
        for c in range(chunks):
            c_lo, c_hi = chunk_bounds(0, B, chunks, c)
            if c_hi <= c_lo:
                continue
            if rank == 0:   # receives first: the peers' chunk c travels while the root walks its own
                recv_into = lands[k % 2] if lands else fulls[k % 2]
                ops = [dist.P2POp(dist.irecv, recv_into[r * B + c_lo:r * B + c_hi], r) for r in range(1, world)]
                pend.works += dist.batch_isend_irecv(ops)
                pend.pieces += [(r * B + c_lo, r * B + c_hi) for r in range(1, world)]
            ev0 = torch.cuda.Event(enable_timing=True)
            ev1 = torch.cuda.Event(enable_timing=True)
            ev0.record(main)
            if csr:
                blk = rw.walk_ex(row_ptr, col_idx, targets[c_lo:c_hi], p, q, L, args.seed, walker_base=walker_base + c_lo,
                                 rng_mode=mode, check=False, out=mine[c_lo:c_hi])
            else:
                blk = walk(targets, walker_base, False)
                if rank == 0:
                    mine.copy_(blk, non_blocking=True)
            ev1.record(main)
            if timed:
                walk_events.append((k, ev0, ev1))
            if rank == 0:
                if pend.ev_first is None:
                    pend.ev_first = ev1
                pend.ev_walked = ev1
            else:
                w = blk if (direct32 or not wire32) else rw.narrow_rows(blk, out=wires[k % 2][c_lo:c_hi])
                if dry:
                    w = w.cpu()
                pend.keep.append(w)
                pend.works += dist.batch_isend_irecv([dist.P2POp(dist.isend, w, 0)])
        return (mine if mine is not None else blk), pend

    walks = None
    # Warm-up calls are where a graph gets prepared (rw.walk keeps the prepared graph of the tensors it has seen:
    # the first call builds what one call is worth, the second node2vec call the full neighbourhood masks); their
    # prepare times are reported beside the steady-state number.
    lib.trw_profile_enable(1)
    cold_prepare_ms = []
    cold = None
    csr_biased = not (kg or elist or windows or uniform)
    if csr_biased:
        cold = {}
        pm = ctypes.c_double()
        if not args.steady:
            # (1) the first call over a graph nobody has prepared: what a one-off caller pays, prepare pass included
            torch.cuda.synchronize()
            t_c = time.perf_counter()
            walks, pend = step_once()
            if pend is not None:
                pend.wait()
            torch.cuda.synchronize()
            cold = dict(first_call_ms=(time.perf_counter() - t_c) * 1e3)
            if lib.trw_profile_prepare_last(ctypes.byref(pm)) == 0:
                cold["first_call_prepare_ms"] = pm.value
            a_, b_, c_ = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
            if lib.trw_profile_last(ctypes.byref(a_), ctypes.byref(b_), ctypes.byref(c_)) == 0:
                cold["first_call_walk_kernel_ms"] = b_.value
        # (2) the graph is walked again and again from here on (K timed steps = K epochs of node2vec walks): prepare it
        # in full once, as a caller who knows that would (rw.prepare); the one-off cost is reported, not hidden
        t_c = time.perf_counter()
        rw.prepare(row_ptr, col_idx, marks={"full": "full", "default": True, "none": False}[args.marks])
        torch.cuda.synchronize()
        cold["prepare_call_ms"] = (time.perf_counter() - t_c) * 1e3
        cold["prepare_marks"] = args.marks
        # where a prepare call's time goes: the allocator (prepared graph + scratch), enqueueing, the kernels
        pt = rw.prepare_timing()
        cold["prepare_alloc_ms"], cold["prepare_launch_ms"] = pt["alloc_ms"], pt["launch_ms"]
        cold["prepare_alloc_bytes"] = pt["prepared_bytes"] + pt["scratch_bytes"]
        if lib.trw_profile_prepare_last(ctypes.byref(pm)) == 0:
            cold["prepare_kernels_ms"] = pm.value
    for _ in range(args.warmup):
        walks = None   # let the allocator hand the same block out again (matters when a result is tens of GB)
        walks, pend = step_once()
        if pend is not None:
            pend.wait()
        pm = ctypes.c_double()
        if lib.trw_profile_prepare_last(ctypes.byref(pm)) == 0:
            cold_prepare_ms.append(pm.value)
    torch.cuda.synchronize()

    # ---- timed region: exactly K steps between barriers -------------------------------------
    # N > 1: the gather of step k rides RCCL's stream while step k+1 walks (the blocks of
    # successive steps are independent); every gather has completed before the closing barrier.
    kernel_ms, prepare_ms, finish_ms = [], [], []
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pending = None
    timed_steps = []
    for _ in range(args.steps):
        walks = None
        walks, work = step_once(timed=True)
        if work is not None:
            timed_steps.append(work)
            if pending is not None:
                pending.wait()
            pending = work
        a, b, c = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        # reading the events of step k waits for that step's kernels; the next step is not queued yet
        # (N > 1 with a gather: the walk calls are timed with events read after the region instead -- nothing may
        # hold the host back between a chunk's walk and the next one's)
        if not gather and lib.trw_profile_last(ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)) == 0:
            prepare_ms.append(a.value)
            kernel_ms.append(b.value)
            finish_ms.append(c.value)
    if pending is not None:
        pending.wait()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    lib.trw_profile_enable(0)

    multi = None
    if gather:
        # walk time of a step = its chunk calls, from events on the launch stream (whole calls: launch checks, the walk
        # kernel, the status seed); max over ranks.  Gather: on the root, from the moment its first own chunk is
        # walked (the peers' first chunks start travelling about then) until the last block is widened in place.
        per_step = {}
        for k, ev0, ev1 in walk_events:
            per_step[k] = per_step.get(k, 0.0) + ev0.elapsed_time(ev1)
        kernel_ms = [per_step[k] for k in sorted(per_step)]
        g_ms = [w.ev_first.elapsed_time(w.ev_done) for w in timed_steps] if rank == 0 else []
        g_tail = [max(0.0, w.ev_walked.elapsed_time(w.ev_done)) for w in timed_steps] if rank == 0 else []
        multi = dict(wire_dtype="uint32" if wire32 else "int64", result_dtype="int32" if result32 else "int64",
                     rows_narrowed_by="walk kernel" if direct32 else ("narrowing pass" if wire32 else None),
                     bytes_per_rank=B * cols * wire_bytes,
                     gather_chunks=chunks, backend=args.backend,
                     gather_ms=sum(g_ms) / max(1, len(g_ms)),
                     gather_after_root_walk_ms=sum(g_tail) / max(1, len(g_tail)),
                     root_ingest_bytes_per_step=(world - 1) * B * cols * wire_bytes)
    t = torch.tensor([elapsed, sum(kernel_ms) / max(1, len(kernel_ms)) if gather else 0.0], dtype=torch.float64,
                     device="cpu" if dry else dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t[0].item())
    if multi is not None:
        multi["walk_kernel_ms"] = float(t[1].item())   # max over ranks of the mean walk time per step

    # ---- device-side status of the last step (validity of what was timed) -------------------
    def as_int64(t):   # narrow rows (uint32 bit patterns in an int32 tensor) as the reference's int64 rows
        return rw.widen_rows(t.contiguous()) if t.dtype == torch.int32 else t
    if not args.steady:
        chk = walk(targets[:1024], walker_base, True)
        assert torch.equal(chk, as_int64(walks[:1024])), "timed output differs from a checked re-run"
    if gather and rank == 0:
        # the gathered blocks are exactly what each rank's global walker indices produce
        for r in range(world):
            blk = walk(targets[:256], r * B, True)
            assert torch.equal(blk.to(full.device), as_int64(full[r * B: r * B + 256])), "gathered block %d is wrong" % r

    # ---- algorithmic bytes of one launch ----------------------------------------
    counters = None
    if windows:
        # per window: ws centre/positive entries read once (8 B) + ws written (8 B) + (ws-1) negatives written (8 B)
        ws_ = cfg["window_size"]
        n_win = B * (L + 1 - ws_ + 1)
        alg_bytes = n_win * (8 * ws_ + 8 * ws_ + 8 * (ws_ - 1))
        kernel_name = "glibc_fill_mod_kernel+windows_gather_kernel"
    elif elist and uniform:
        # uniform: per step 16 B index pair + 16 B edge row + 8 B stored
        alg_bytes = B * (16 + 40 * L)
        kernel_name = "elist_walk_kernel"
    elif elist:
        # biased: the counting variant of the kernel through the C ABI (same walks), then bytes the algorithm as run
        # must move: per walker 8 B target + 8 B index pair; 8 B per stored column; 48 B read per entry gathered (every
        # accepted offset + the candidates the masks could not decide); 16 B per table slot read
        lib.trw_walk_edge_list_workspace_bytes.restype = ctypes.c_size_t
        lib.trw_walk_edge_list_workspace_bytes.argtypes = [ctypes.c_int64] * 3 + [ctypes.c_int32, ctypes.c_int64, ctypes.c_int32]
        nbytes = lib.trw_walk_edge_list_workspace_bytes(nnz, n, B, L, pad, mode)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        out = torch.empty((B, L + 1), dtype=torch.int64, device=dev)
        stream = torch.cuda.current_stream().cuda_stream
        lib.trw_profile_enable(2)
        rc = lib.trw_walk_edge_list_hip(el_edges.data_ptr(), nnz, el_index.data_ptr(), n, targets.data_ptr(), B, walker_base,
                                        p, q, L, args.seed, pad, 1, mode, out.data_ptr(), ws.data_ptr(), nbytes, stream)
        assert rc == 0, lib.trw_last_error()
        c = (ctypes.c_uint64 * 4)()
        assert lib.trw_walk_counters(ws.data_ptr(), stream, c) == 0
        stw = (ctypes.c_int64 * 4)()
        assert lib.trw_walk_status(ws.data_ptr(), stream, stw) == 0
        lib.trw_profile_enable(0)
        if not torch.equal(out, as_int64(walks)):
            bad = (out != as_int64(walks)).any(1).nonzero().flatten()
            raise AssertionError("counting variant of the kernel produced different walks: %d rows differ, first %s"
                                 % (bad.numel(), bad[:8].tolist()))
        trials, fetches, tests, probes = (int(x) for x in c)
        if int(stw[3]) == 5:
            kernel_name = "elist_pool_kernel"
            counters = dict(trials=trials, entry_gathers=fetches, candidates_tested=tests, table_slots=probes)
            alg_bytes = B * 16 + 8 * B * (L + 1) + 48 * fetches + 16 * probes
        else:   # the lane-per-walker kernel loads every candidate (4 B) and an 8-byte index pair per step
            kernel_name = "elist_walk_kernel"
            alg_bytes = B * (16 + 40 * L)
        del out, ws
    elif kg:
        # per LIVE hop (not yet on the padding id / a dead end): 16 B index pair + 16 B (relation, tail);
        # per hop 16 B stored; per walker 8 B target + 8 B column 0.  Hops after a dead end read nothing.
        live_hops = int((walks[:, 1::2] != pad).sum())
        counters = dict(live_hops=live_hops, hops=B * L)
        alg_bytes = B * 16 + 32 * live_hops + 16 * B * L
        kernel_name = "triples_walker_kernel"
    elif fused_ws:
        # the walk's reads (8 B target + 8 B row pair per walker, one 16-byte entry gathered + 16 B of row32 per step
        # as in the plain uniform walk's count) + every window's centre, positives and negatives written once
        n_win = B * (L + 1 - fused_ws + 1)
        alg_bytes = B * (16 + 24 * L) + n_win * 8 * (1 + 2 * (fused_ws - 1))
        kernel_name = "uniform_glibc_rows_kernel<fat,windows> || glibc_fill_mod_kernel"
        counters = dict(windows=n_win, window_size=fused_ws,
                        fused=bool(lib.trw_walk_windows_fused(n, nnz, B, L, fused_ws, 1)))
    elif uniform:
        alg_bytes = B * (16 + 32 * L)
        # which kernel family ran: one more (untimed) call through the C ABI over a graph prepared exactly as the timed
        # calls' was (the cache may have added the 16-byte entries over the warm-up calls), then its status block
        cached = [i for i in rw.cache_info() if i[0] == n and i[1] == nnz]
        what = int(cached[0][2]) if cached else int(lib.trw_prepare_plan(n, nnz, B, L, p, q, mode))
        out = torch.empty((B, L + 1), dtype=torch.int64, device=dev)
        stream = torch.cuda.current_stream().cuda_stream
        if what >= 0:
            rw.clear_cache()
            torch.cuda.empty_cache()
            pbytes = lib.trw_prepare_csr_bytes(n, nnz, what)
            blob = torch.empty(pbytes, dtype=torch.uint8, device=dev)
            rc = lib.trw_prepare_csr_hip(row_ptr.data_ptr(), n, col_idx.data_ptr(), nnz, what, blob.data_ptr(), pbytes, stream)
            assert rc == 0, lib.trw_last_error()
            nbytes = lib.trw_walk_csr_prepared_workspace_bytes(B, L, p, q, mode)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            rc = lib.trw_walk_csr_prepared_hip(blob.data_ptr(), pbytes, what, row_ptr.data_ptr(), n, col_idx.data_ptr(), nnz,
                                               targets.data_ptr(), B, walker_base, p, q, L, args.seed, mode, out.data_ptr(),
                                               ws.data_ptr(), nbytes, stream)
        else:
            nbytes = lib.trw_walk_csr_workspace_bytes(n, nnz, B, L, p, q, mode)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            rc = lib.trw_walk_csr_hip(row_ptr.data_ptr(), n, col_idx.data_ptr(), nnz, targets.data_ptr(), B,
                                      walker_base, p, q, L, args.seed, mode, out.data_ptr(), ws.data_ptr(), nbytes,
                                      stream)
        assert rc == 0, lib.trw_last_error()
        counters = dict(prepared_what=what)
        stw = (ctypes.c_int64 * 4)()
        assert lib.trw_walk_status(ws.data_ptr(), stream, stw) == 0 and stw[0] == 0
        assert torch.equal(out, as_int64(walks)), "direct C-ABI call produced different walks"
        kernel_name = {0: "uniform_glibc_wide_kernel", 1: "uniform_glibc_rows_kernel<false>",
                       3: "uniform_glibc_rows_kernel<true>"}[int(stw[3])]
        blob = None
        del out, ws
    else:
        # the counting variant of the kernel over a graph prepared exactly as the timed calls' was (their cached
        # prepared graph is dropped first: two of them need not fit side by side)
        cached = [i for i in rw.cache_info() if i[0] == n and i[1] == nnz]
        what = int(cached[0][2]) if cached else int(lib.trw_prepare_plan(n, nnz, B, L, p, q, mode))
        chk = None
        rw.clear_cache()
        torch.cuda.empty_cache()
        stream = torch.cuda.current_stream().cuda_stream
        pbytes = lib.trw_prepare_csr_bytes(n, nnz, what)
        blob = torch.empty(pbytes, dtype=torch.uint8, device=dev)
        rc = lib.trw_prepare_csr_hip(row_ptr.data_ptr(), n, col_idx.data_ptr(), nnz, what, blob.data_ptr(), pbytes, stream)
        assert rc == 0, lib.trw_last_error()
        lib.trw_profile_enable(2)
        nbytes = lib.trw_walk_csr_prepared_workspace_bytes(B, L, p, q, mode)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        out = torch.empty((B, L + 1), dtype=torch.int64, device=dev)
        rc = lib.trw_walk_csr_prepared_hip(blob.data_ptr(), pbytes, what, row_ptr.data_ptr(), n, col_idx.data_ptr(), nnz,
                                           targets.data_ptr(), B, walker_base, p, q, L, args.seed, mode, out.data_ptr(),
                                           ws.data_ptr(), nbytes, stream)
        assert rc == 0, lib.trw_last_error()
        c = (ctypes.c_uint64 * 4)()
        assert lib.trw_walk_counters(ws.data_ptr(), stream, c) == 0
        stw = (ctypes.c_int64 * 4)()
        assert lib.trw_walk_status(ws.data_ptr(), stream, stw) == 0
        family = int(stw[3])
        kernel_name = {0: "biased_walker_wide_kernel", 1: "biased_walker_kernel", 2: "biased_hash_kernel",
                       4: "biased_mask_kernel", 5: "biased_pool_kernel"}[family]
        lib.trw_profile_enable(0)
        if not torch.equal(out, as_int64(walks)):
            bad = (out != as_int64(walks)).any(1).nonzero().flatten()
            raise AssertionError("counting variant of the kernel produced different walks: %d rows differ, first %s"
                                 % (bad.numel(), bad[:8].tolist()))
        trials, fetches, tests, probes = (int(x) for x in c)
        # the prepared graph proper (what the walks read) and the marking passes' scratch, which the torch layer gives back
        # to the allocator when the prepare call returns
        keep_bytes = int(lib.trw_prepare_csr_persistent_bytes(n, nnz, what))
        counters = dict(trials=trials, candidate_loads=fetches, membership_tests=tests, table_or_bisect_entries=probes,
                        prepared_what=what, prepared_bytes=keep_bytes,
                        prepared_bytes_per_edge=round(keep_bytes / max(1, int(nnz)), 2),
                        prepare_scratch_bytes=int(lib.trw_prepare_csr_scratch_bytes(n, nnz, what)))
        if family in (4, 5):
            # biased_pool_kernel / biased_mask_kernel, bytes the algorithm as run must move: per walker 8 B target + 8 B row32 pair; per
            # column 8 B stored; 32 B per fat entry gathered (every accepted offset, plus candidates the table is
            # asked about); 32 B per probe of the edge set (a probe reads a block of four 8-byte keys, an aligned half of
            # one line).  (Rounds 2-4 counted 16 B per probe, from when a probe was a pair of 16-byte slots: that figure is
            # kept beside it as alg_bytes_r04_convention.)
            alg_bytes = B * 16 + 8 * B * (L + 1) + 32 * fetches + 32 * probes
            counters["alg_bytes_r04_convention"] = B * 16 + 8 * B * (L + 1) + 32 * fetches + 16 * probes
        else:
            # int64-API-layout bytes (with this kernel's own counts): per walker 8 B target + 8 B column 0
            # + 32 B first step; per later step 16 B row_ptr pair + 8 B stored column; 8 B per candidate entry loaded
            # and per table/col_idx entry a lookup reads.
            alg_bytes = B * 16 + B * 32 + (L - 1) * B * (16 + 8) + 8 * fetches + 8 * probes
        # (The reference's own access pattern -- a candidate load and a linear row scan on every trial -- counts
        # many times more: roofline.reference_accounting.)
        del out, ws, blob

    steps_per_launch = B * L
    k_ms = sum(kernel_ms) / max(1, len(kernel_ms))
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0

    # roofline.traffic / request_rate come from a separate PMC pass of this same command (scripts/profile_gpu.sh ->
    # profiles/pmc_traffic.json): quoted only while that pass was collected from the kernel sources that are running
    # now (sha256 over csrc/), null otherwise -- a stale figure is worse than none.
    traffic, request_rate, traffic_source = None, None, None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        import hashlib
        h = hashlib.sha256()
        cdir = os.path.join(ROOT, "torch_random_walk_amd", "csrc")
        for name in sorted(os.listdir(cdir)):
            if name.endswith((".hip", ".hpp", ".cpp")) and name != "torch_rw_native.cpp":
                with open(os.path.join(cdir, name), "rb") as f:
                    h.update(f.read())
        now_sha = h.hexdigest()
        with open(pmc_path) as f:
            ent = json.load(f).get(args.workload, {})
        fresh = bool(ent) and ent.get("sources_sha256") == now_sha and ent.get("kernel", "") in kernel_name
        traffic_source = dict(file="profiles/pmc_traffic.json", tag=ent.get("tag"), kernel=ent.get("kernel"),
                              sources_sha256=ent.get("sources_sha256"), current_sources_sha256=now_sha,
                              matches_current_sources=fresh)
        if fresh:
            traffic = ent.get("hbm_bytes_per_launch")
            misses = ent.get("TCC_MISS")
            # The gather kernels are bound by the rate of L2-miss requests the memory system sustains
            # (scripts/machine_probe.hip measures it on the box: the same for 4-, 8- and 16-byte gathers), not by
            # bytes: quote the kernel's miss rate against that measured ceiling next to the HBM roofline.
            ceiling, copy_gbs = None, None
            import glob
            probe_path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_machine_probe.txt")))[-1]
            with open(probe_path) as f:
                for ln in f:
                    tok = ln.split()
                    if ln.startswith("stream_copy"):
                        copy_gbs = float(ln.split("per copy,")[1].split()[0])
                    elif len(tok) >= 5 and tok[1] == "MB" and int(tok[0]) >= 4096:
                        best = max(float(x) for x in tok[2:])
                        ceiling = best if ceiling is None else min(ceiling, best)  # the largest footprints
            if misses and ceiling and k_ms > 0:
                request_rate = dict(l2_misses_per_launch=misses, achieved_per_s=misses / (k_ms * 1e-3),
                                    ceiling_per_s=ceiling, frac=misses / (k_ms * 1e-3) / ceiling,
                                    measured_stream_copy_GBps=copy_gbs,
                                    source="profiles/pmc_traffic.json (TCC_MISS) / profiles/%s "
                                           "(dependent random gathers over >= 4 GB, 8 waves per SIMD)"
                                           % os.path.basename(probe_path))
    except Exception as e:   # noqa: BLE001 -- the PMC figures are optional
        traffic_source = dict(file="profiles/pmc_traffic.json", error=repr(e))

    # ---- CPU baseline: the reference's algorithm on a bounded sample (rank 0, N = 1 only) ------
    cpu = None
    reference_accounting = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O
        g = torch.Generator().manual_seed(12345)
        perm = torch.randperm(B, generator=g)
        tg_cpu = targets.cpu()
        if windows:
            ww_h = win_walks.cpu().numpy()

            def cpu_run(tg_h):
                import time as _t
                t_ = _t.perf_counter()
                O.to_windows(ww_h[:len(tg_h)], cfg["window_size"], n, args.seed, rng_mode=O.RNG_LIBC)
                return _t.perf_counter() - t_
        elif elist:
            ee_h, ei_h = el_edges.cpu().numpy(), el_index.cpu().numpy()

            def cpu_run(tg_h):
                import time as _t
                t_ = _t.perf_counter()
                O.walk_edge_list(ee_h, ei_h, tg_h, p, q, L, args.seed, pad, restart=True, rng_mode=O.RNG_LIBC)
                return _t.perf_counter() - t_
        elif kg:
            ts_h, ix_h = triples.cpu().numpy(), rt_index.cpu().numpy()

            def cpu_run(tg_h):
                return O.walk_triples(ts_h, ix_h, tg_h, L, pad, args.seed, rng_mode=O.RNG_LIBC, timed=True)[1]
        elif fused_ws:
            rp_h, ci_h = row_ptr.cpu().numpy(), col_idx.cpu().numpy()

            def cpu_run(tg_h):   # the reference's two calls: walk, then to_windows over its rows
                import time as _t
                w_h, dt_w = O.walk_timed(rp_h, ci_h, tg_h, p, q, L, args.seed, rng_mode=O.RNG_LIBC)
                t_ = _t.perf_counter()
                O.to_windows(w_h, fused_ws, n, args.seed, rng_mode=O.RNG_LIBC)
                return dt_w + _t.perf_counter() - t_
        else:
            rp_h, ci_h = row_ptr.cpu().numpy(), col_idx.cpu().numpy()

            def cpu_run(tg_h):
                return O.walk_timed(rp_h, ci_h, tg_h, p, q, L, args.seed, rng_mode=O.RNG_LIBC)[1]
        probe = 5000 if (uniform or kg or windows) else 500
        dt = cpu_run(tg_cpu[perm[:probe]].numpy())
        rate = probe * L / dt
        sample = int(min(B, max(probe, args.cpu_seconds * rate / L)))
        dt = cpu_run(tg_cpu[perm[:sample]].numpy())
        cpu = dict(value=sample * L / dt, unit="walk-steps/s", cores=1, kind="port",
                   sample="first %d start nodes of a seeded permutation of the batch, %d steps, %.1f s, "
                          "oracle libc srand/rand single thread (the reference's csrc/cpu algorithm; its "
                          "multi-threaded run is slower and non-deterministic, SURVEY 0.2)" % (sample, sample * L, dt))
        if not (windows or elist or kg or fused_ws):
            # the reference's multi-threaded policy beside the single thread (all host threads on the
            # one locked libc rand(); non-deterministic output), a quarter of the time budget; the faster one is
            # the baseline.
            nthreads = min(os.cpu_count() or 1, 16)  # a GPU box's CPU share
            probe_mt = max(2 * 32768 // (1 if uniform else L + 1), 2000)
            _, dt_mt, used = O.walk_timed_mt(rp_h, ci_h, tg_cpu[perm[:probe_mt]].numpy(), p, q, L, args.seed, nthreads)
            rate_mt = probe_mt * L / dt_mt
            sample_mt = int(min(B, max(probe_mt, 0.25 * args.cpu_seconds * rate_mt / L)))
            _, dt_mt, used = O.walk_timed_mt(rp_h, ci_h, tg_cpu[perm[:sample_mt]].numpy(), p, q, L, args.seed, nthreads)
            rate_mt = sample_mt * L / dt_mt
            cpu["single_thread"] = cpu["value"]
            cpu["multi_thread"] = dict(value=rate_mt, threads=used, host_threads=nthreads,
                                       sample="%d start nodes, %.1f s, at::parallel_for chunking, every thread on "
                                              "libc's locked rand()" % (sample_mt, dt_mt))
            if rate_mt > cpu["value"]:
                cpu["value"], cpu["cores"] = rate_mt, used
        m = min(B, 2000)
        if windows:
            ref_n = O.to_windows(ww_h[:m], cfg["window_size"], n, args.seed)[2].reshape(m, -1)
            assert (ref_n == walks[:m].cpu().numpy()).all(), "GPU negatives differ from the oracle"
        elif elist:
            ref_w = O.walk_edge_list(ee_h, ei_h, tg_cpu[:m].numpy(), p, q, L, args.seed, pad, restart=True,
                                     rng_mode=O.RNG_WALKER, walker_base=walker_base)
            assert (ref_w == walks[:m].cpu().numpy()).all(), "GPU rows differ from the oracle"
        elif kg:
            ref_w = O.walk_triples(ts_h, ix_h, tg_cpu[:m].numpy(), L, pad, args.seed, rng_mode=O.RNG_WALKER,
                                   walker_base=walker_base)
            assert (ref_w == walks[:m].cpu().numpy()).all(), "GPU rows differ from the oracle"
        elif fused_ws:
            assert walker_base == 0
            ref_w = O.walk(rp_h, ci_h, tg_cpu[:m].numpy(), p, q, L, args.seed, rng_mode=O.RNG_GLIBC_INDEXED)
            ref_p = O.to_windows(ref_w, fused_ws, n, args.seed)[1].reshape(m, -1)
            assert (ref_p == walks[:m].cpu().numpy()).all(), "GPU windows differ from the oracle"
        elif uniform:
            ref_w = O.walk(rp_h, ci_h, tg_cpu[:m].numpy(), p, q, L, args.seed, rng_mode=O.RNG_GLIBC_INDEXED,
                           walker_base=walker_base)
            assert (ref_w == as_int64(walks[:m]).cpu().numpy()).all(), "GPU rows differ from the oracle"
        else:
            # SURVEY 8(d) accounting: what the REFERENCE's loop reads for the same walks (a candidate on every
            # trial, is_neighbor on every candidate != t), from the oracle's counters on walkers 0..m-1 -- the
            # same walks the GPU produced for those rows -- scaled to the batch.
            import numpy as np
            ref_w, cnt = O.walk(rp_h, ci_h, tg_cpu[:m].numpy(), p, q, L, args.seed, rng_mode=O.RNG_WALKER,
                                walker_base=walker_base, return_counters=True)
            assert (ref_w == as_int64(walks[:m]).cpu().numpy()).all(), "GPU rows differ from the oracle"
            scale = B / m
            deg_h = (rp_h[1:] - rp_h[:-1])
            t_nodes = ref_w[:, :-2].reshape(-1)            # the `t` of every step >= 2
            log2deg = np.ceil(np.log2(np.maximum(deg_h[t_nodes], 2))).mean()
            trials_r, member_r, scan_r = cnt["trials_total"] * scale, cnt["membership_tests"] * scale, \
                cnt["scan_entries_total"] * scale
            base_r = B * 16 + B * 32 + 8 * B * (L - 1) + 24 * trials_r + 16 * member_r
            reference_accounting = dict(
                sample_walkers=m,
                trials=trials_r, membership_tests=member_r, linear_scan_entries=scan_r,
                bytes_linear_scan=base_r + 8 * scan_r,
                bytes_binary_search=base_r + 8 * member_r * float(log2deg),
                note="SURVEY 8(d) formula with the reference's access pattern; NOT used for roofline.achieved, "
                     "which counts only what this kernel's algorithm has to read")

    # What the prepared graph costs a caller who does not walk it for ever: after how many epochs
    # the one-off full prepare has paid for itself against walking over what the first call builds by itself, and the
    # rate over 10 epochs with the prepare call inside.
    amortised = None
    if cold and all(k in cold for k in ("first_call_prepare_ms", "first_call_walk_kernel_ms", "prepare_call_ms")):
        steady = elapsed / args.steps * 1e3
        gain = cold["first_call_walk_kernel_ms"] - steady
        extra = cold["prepare_call_ms"] - cold["first_call_prepare_ms"]
        amortised = dict(
            epochs_to_break_even=(extra / gain) if gain > 0 else None,
            value_incl_prepare_at_10_epochs=10 * world * B * L / ((cold["prepare_call_ms"] + 10 * steady) * 1e-3),
            value_one_shot_call=world * B * L / (cold["first_call_ms"] * 1e-3),
            note="break-even: (full prepare - the first call's own prepare) / (first call's walk - steady-state step); "
                 "one_shot = a single call over a graph nobody prepared, everything included")
    if rank == 0:
        total_steps = world * B * L * args.steps
        line = {
            "metric": "walk-steps/s",
            "value": total_steps / elapsed,
            "unit": "walk-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int64",
            "data": "synthetic",
            "walks_per_s": world * B * args.steps / elapsed,
            "config": {
                "workload": args.workload,
                "graph": ("kg entities=%d triples=%d relations=%d graph_seed=%d" % (n, nnz, cfg["relations"],
                                                                                      args.graph_seed)) if kg else
                         ("random walks [%d, %d] over %d nodes, window_size=%d" % (B, L + 1, n, cfg["window_size"]))
                         if windows else
                         "%s nodes=%d directed_edges=%d (+ring) nnz=%d graph_seed=%d" % (
                    cfg["graph"], n, cfg["edges"], nnz, args.graph_seed),
                "p": p, "q": q, "walk_length": L, "seed": args.seed,
                "walkers_per_gpu": B, "global_walkers": world * B,
                "rng": "per-walker philox4x32-10" if mode == rw.RNG_WALKER else "glibc-indexed (csrc/cpu-exact)",
                "partition": "start nodes block-sharded, CSR replicated",
                "collective": ("one RCCL gather to rank 0 per step (groups of peer->root transfers, %d row chunks, %s on "
                               "the wire, widened into the destination slices), inside the timed region, overlapped "
                               "with the rest of the step's walk and the next step's" % (
                                   chunks, "uint32" if wire32 else "int64")) if gather else "none",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": kernel_name,
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_source,
                "algorithmic_bytes_per_launch": alg_bytes,
                "units_per_launch": steps_per_launch,
                "kernel_ms": k_ms,
                "prepare_ms": sum(prepare_ms) / max(1, len(prepare_ms)),
                "finish_ms": sum(finish_ms) / max(1, len(finish_ms)),
                "cold_prepare_ms": cold_prepare_ms,   # prepare passes of the warm-up calls, if any
                "cold": cold,                         # CSR biased walk: first call over an unprepared graph; the one-off full prepare
                "amortised": amortised,
                "kernel_steps_per_s": steps_per_launch / (k_ms * 1e-3) if k_ms > 0 else None,
                "counters": counters,
                "request_rate": request_rate,
                "reference_accounting": reference_accounting,
            },
            "cpu_baseline": cpu,
            "multi_gpu": multi,
            "graph_build_s": t_graph,
            "peak_hbm_gb": torch.cuda.max_memory_allocated() / 1e9,   # rank 0: graph + workspace + results (+ generator temporaries)
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
