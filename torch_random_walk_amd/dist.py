"""Start-node batches sharded over the GPUs of one node.

The reference is single-device (csrc/cuda/rw_cuda.cu:198); this is the one
multi-GPU step the path has.  The CSR is replicated on every rank, the start
nodes are block-partitioned, each rank walks its block with the block's global
offset as ``walker_base`` (so the assembled result is bit-identical to the
single-GPU call made with the same explicit RNG mode), and ONE collective
gathers the row blocks.  No collective is needed by the walk itself.
"""
import torch
import torch.distributed as dist

from . import rw


def shard_bounds(n, world_size, rank):
    """Block partition: rank r owns [r*B, min(n, (r+1)*B)), B = ceil(n / world_size)."""
    block = (n + world_size - 1) // world_size if world_size > 0 else n
    lo = min(n, rank * block)
    hi = min(n, lo + block)
    return lo, hi, block


def chunk_bounds(lo, hi, chunks, c):
    """Row sub-block c of the block [lo, hi) cut into `chunks` pieces of ceil((hi-lo)/chunks) rows (the last ones
    may be short or empty).  Sender and root derive the same cuts from the shard bounds alone."""
    size = (hi - lo + chunks - 1) // chunks if chunks > 0 else hi - lo
    c_lo = min(hi, lo + c * size)
    return c_lo, min(hi, c_lo + size)


def wire_dtype_for(n_nodes, wire="auto"):
    """What a row block travels as: "uint32" (half the bytes per xGMI link; node ids are < n_nodes) while
    n_nodes <= 2^32, "int64" otherwise or on request."""
    if wire == "auto":
        return "uint32" if n_nodes <= (1 << 32) else "int64"
    if wire == "uint32" and n_nodes > (1 << 32):
        raise ValueError("node ids of a graph with more than 2^32 nodes do not fit the uint32 wire format")
    if wire not in ("uint32", "int64"):
        raise ValueError("wire must be 'auto', 'uint32' or 'int64'")
    return wire


def _host_staged(group, tensor):
    """gloo moves host memory only (its send/recv and all_gather refuse device tensors): device blocks
    are staged through the host then.  RCCL (backend "nccl") takes the device tensors as they are."""
    return tensor.is_cuda and dist.get_backend(group) == "gloo"


def walk_sharded(row_ptr, col_idx, target_nodes, p, q, walk_length, seed, group=None, gather="root",
                 rng_mode=None, walk_fn=None, wire="auto", chunks=1, result_dtype=torch.int64):
    """Walk ``target_nodes`` (the full batch, same on every rank) across the ranks of ``group``.

    gather: "root"  -> rank 0 returns the full ``[N, L+1]`` tensor, other ranks return their block
            "all"   -> every rank returns the full tensor (all_gather)
            "none"  -> every rank returns its own block only
    rng_mode: default = GLIBC for uniform walks (indexed jump-ahead shards exactly),
              WALKER for biased walks.  The result equals ``rw.walk_ex(..., rng_mode=<that mode>)`` on
              one GPU for any world size.  It is NOT what ``rw.walk`` returns under RNG_AUTO for a small
              biased batch (N <= 32768/(L+1)): AUTO replays the reference's one serial glibc chain there,
              and a serial chain cannot be sharded -- asking for RNG_GLIBC on a biased walk with more
              than one rank raises.
    wire:   "auto" | "uint32" | "int64" -- what the blocks travel as (``wire_dtype_for``): uint32 halves the bytes
            on every link; the blocks are narrowed by a kernel of the library before they leave and widened into
            the int64 result where they land.  The result is the same tensor either way.
    chunks: gather="root": a rank walks its block in this many row sub-blocks and posts each one's transfer as
            soon as its walk is enqueued, so the gather overlaps the rest of the same call's walk instead of
            starting when the whole block is done.
    result_dtype: ``torch.int64`` (the reference's rows) or ``torch.int32`` = NARROW rows end to end: what is returned
            -- the gathered tensor and every rank's own block -- holds the node ids as uint32 bit patterns in an int32
            tensor (``rw.widen_rows`` turns them into int64).  Every rank's walk kernel writes them itself
            (``rw.walk_ex(out=<int32 tensor>)``), the peers' blocks land in the result, nothing is narrowed or widened
            on the way: half the bytes to write, to send and to keep.  Needs node ids below 2^32 (the uint32 wire).
    walk_fn: override of ``rw.walk_ex`` (the CPU tests plug the oracle in here).
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = int(target_nodes.numel())
    lo, hi, block = shard_bounds(n, world, rank)
    uniform = (p == 1.0 and q == 1.0)
    if rng_mode is None or rng_mode == rw.RNG_AUTO:
        rng_mode = rw.RNG_GLIBC if uniform else rw.RNG_WALKER
    if world > 1 and rng_mode == rw.RNG_GLIBC and not uniform:
        raise ValueError("RNG_GLIBC biased walks replay one serial stream (walker i's draws depend on every "
                         "earlier walker's): they cannot be sharded; use RNG_WALKER")
    if gather not in ("root", "all", "none"):
        raise ValueError("gather must be 'root', 'all' or 'none'")
    narrow = wire_dtype_for(int(row_ptr.numel()) - 1, wire) == "uint32"
    if result_dtype not in (torch.int64, torch.int32):
        raise ValueError("result_dtype must be torch.int64 or torch.int32")
    res32 = result_dtype == torch.int32
    if res32 and not narrow:
        raise ValueError("result_dtype=torch.int32 holds node ids below 2^32 and travels as uint32: wire must not be 'int64'")
    chunks = max(1, int(chunks)) if (world > 1 and gather == "root") else 1
    cols = walk_length + 1
    dev = target_nodes.device
    staged = world > 1 and _host_staged(group, target_nodes)

    fault = []   # first exception of this rank's walks: remembered, raised only once every transfer is done

    def walk_into(dst, c_lo, c_hi):
        """rows [c_lo, c_hi) of the batch, walked into `dst` (a [c_hi-c_lo, cols] slice of a result tensor).
        NEVER raises: a rank that left the chunk loop half way would strand its peers in their receives (the root in
        irecv.wait() for the chunks that were never sent, or the peers' isends without a match) -- ADVICE r03.  An
        exception -- an argument error, or a data fault of an earlier deferred-check call surfacing at this call's
        entry -- is kept for `settle`; the rank goes on and ships whatever its buffer holds, and `settle` makes sure
        nobody takes that for a result."""
        try:
            if walk_fn is None:
                # (deferred status check: nothing holds the host back between a chunk's walk and its transfer)
                rw.walk_ex(row_ptr, col_idx, target_nodes[c_lo:c_hi], p, q, walk_length, seed, walker_base=c_lo,
                           rng_mode=rng_mode, check="deferred", out=dst)
            else:
                rows = walk_fn(row_ptr, col_idx, target_nodes[c_lo:c_hi], p, q, walk_length, seed,
                               walker_base=c_lo, rng_mode=rng_mode)
                dst.copy_(rw.narrow_rows(rows.contiguous()) if res32 else rows)
        except Exception as e:  # noqa: BLE001 -- see above: kept, raised by settle()
            if not fault:
                fault.append(e)

    def settle(result, collective=True):
        """A block with a data fault (degree-0 node, id out of range) must not pass for a result -- on ANY rank: the
        rows of a faulty block have already travelled when its owner finds out.  Every rank waits for the status
        read-backs of its own walks, the ranks agree on "did anyone fault" (one MAX all_reduce of a flag; every rank
        reaches it, whatever happened before), and all of them raise: the owner its own error, the others a
        RuntimeError that names the faulty ranks.  ``gather="none"`` has no collective and stays
        local."""
        if walk_fn is None and target_nodes.is_cuda:
            try:
                rw.check_status(wait=True)
            except RuntimeError as e:
                if not fault:
                    fault.append(e)
        if world > 1 and collective:
            flags = torch.zeros(world, dtype=torch.int32, device="cpu" if (staged or not target_nodes.is_cuda) else dev)
            if fault:
                flags[rank] = 1
            dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=group)
            bad = [r for r, f in enumerate(flags.tolist()) if f]
            if bad and not fault:
                raise RuntimeError("torch_rw.walk_sharded: the walk of rank(s) %s reported a fault; the gathered rows "
                                   "are not a result" % ", ".join(str(r) for r in bad))
        if fault:
            raise fault[0]
        return result

    def to_wire(rows):
        """what travels for a block of rows: the rows themselves, or their uint32 form; on the host under gloo"""
        w = rw.narrow_rows(rows) if (narrow and not res32) else rows   # (res32: the rows ARE the wire form)
        return w.cpu() if staged else w

    root_gather = world > 1 and gather == "root"
    if root_gather and rank == 0:
        # the root walks straight into its slice of the result: no copy of its own block
        full = torch.empty((world * block, cols), dtype=result_dtype, device=dev)
        mine = full[lo:hi]
    else:
        full = None
        mine = torch.empty((hi - lo, cols), dtype=result_dtype, device=dev)

    if not root_gather:
        walk_into(mine, lo, hi)
        if world == 1 or gather == "none":
            return settle(mine, collective=False)
        # gather == "all": equal-sized pieces for the collective; the pad rows are dropped below
        w = to_wire(mine)
        if hi - lo < block:
            padded = torch.zeros((block, cols), dtype=w.dtype, device=w.device)
            padded[: hi - lo] = w
            w = padded
        got = torch.empty((world * block, cols), dtype=w.dtype, device=w.device)
        dist.all_gather_into_tensor(got, w.contiguous(), group=group)
        got = got[:n]
        if staged:
            got = got.to(dev)
        return settle(rw.widen_rows(got.contiguous()) if (narrow and not res32) else got)

    # gather == "root": a gather on xGMI is world-1 independent peer -> root transfers; issued as groups of
    # point-to-point operations they land directly in the root's buffers, chunk by chunk
    if rank != 0:
        dst = 0 if group is None else dist.get_global_rank(group, 0)
        works, keep = [], []
        for c in range(chunks):
            c_lo, c_hi = chunk_bounds(lo, hi, chunks, c)
            if c_hi <= c_lo:
                continue
            walk_into(mine[c_lo - lo:c_hi - lo], c_lo, c_hi)
            w = to_wire(mine[c_lo - lo:c_hi - lo]).contiguous()
            keep.append(w)   # (alive until the transfer is done)
            works += dist.batch_isend_irecv([dist.P2POp(dist.isend, w, dst, group)])
        for w in works:
            w.wait()
        return settle(mine)

    wire_dtype = torch.int32 if narrow else torch.int64
    land = None
    if (narrow and not res32) or staged:   # where the peers' blocks arrive before they are widened / copied into `full`
        land = torch.empty((world * block, cols), dtype=wire_dtype, device="cpu" if staged else dev)
    pending = []
    for c in range(chunks):
        ops, pieces = [], []
        for r in range(1, world):
            r_lo, r_hi, _ = shard_bounds(n, world, r)
            c_lo, c_hi = chunk_bounds(r_lo, r_hi, chunks, c)
            if c_hi <= c_lo:
                continue
            src = r if group is None else dist.get_global_rank(group, r)
            ops.append(dist.P2POp(dist.irecv, (land if land is not None else full)[c_lo:c_hi], src, group))
            pieces.append((c_lo, c_hi))
        # receives first, then this chunk of the root's own walk: the peers' chunk c travels while the root walks
        works = dist.batch_isend_irecv(ops) if ops else []
        c_lo, c_hi = chunk_bounds(lo, hi, chunks, c)
        if c_hi > c_lo:
            walk_into(full[c_lo:c_hi], c_lo, c_hi)
        pending.append((works, pieces))
    for works, pieces in pending:
        for w in works:
            w.wait()
        if land is None:
            continue
        for c_lo, c_hi in pieces:
            got = land[c_lo:c_hi].to(dev) if staged else land[c_lo:c_hi]
            if narrow and not res32:
                rw.widen_rows(got, out=full[c_lo:c_hi])
            else:
                full[c_lo:c_hi].copy_(got)
    return settle(full[:n])
