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


def _host_staged(group, tensor):
    """gloo moves host memory only (its send/recv and all_gather refuse device tensors): device blocks
    are staged through the host then.  RCCL (backend "nccl") takes the device tensors as they are."""
    return tensor.is_cuda and dist.get_backend(group) == "gloo"


def walk_sharded(row_ptr, col_idx, target_nodes, p, q, walk_length, seed, group=None, gather="root",
                 rng_mode=None, walk_fn=None):
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
    fn = walk_fn or rw.walk_ex
    cols = walk_length + 1
    full = None
    staged = world > 1 and _host_staged(group, target_nodes)
    if world > 1 and gather == "root" and rank == 0 and walk_fn is None and not staged:
        # the root walks straight into its slice of the result: no copy of its own block
        full = torch.empty((world * block, cols), dtype=torch.int64, device=target_nodes.device)
        mine = fn(row_ptr, col_idx, target_nodes[lo:hi], p, q, walk_length, seed, walker_base=lo, rng_mode=rng_mode,
                  out=full[lo:hi])
    else:
        mine = fn(row_ptr, col_idx, target_nodes[lo:hi], p, q, walk_length, seed, walker_base=lo, rng_mode=rng_mode)
    if world == 1 or gather == "none":
        return mine
    wire = mine.cpu() if staged else mine            # what travels
    padded = wire
    if hi - lo < block:  # equal-sized pieces for the collective; the pad rows are dropped below
        padded = torch.zeros((block, cols), dtype=wire.dtype, device=wire.device)
        padded[: hi - lo] = wire
    if gather == "all":
        got = torch.empty((world * block, cols), dtype=wire.dtype, device=wire.device)
        dist.all_gather_into_tensor(got, padded.contiguous(), group=group)
        return got[:n].to(mine.device) if staged else got[:n]
    if gather == "root":
        # a gather on xGMI is world-1 independent peer -> root transfers; issued as one group of
        # point-to-point operations they land directly in the root's result, no staging copy
        if rank == 0:
            if full is None:
                full = torch.empty((world * block, cols), dtype=mine.dtype, device=mine.device)
                full[: hi - lo] = mine
            land = torch.empty((world * block, cols), dtype=wire.dtype) if staged else full
            ops = []
            for r in range(1, world):
                r_lo, r_hi, _ = shard_bounds(n, world, r)
                if r_hi > r_lo:
                    src = r if group is None else dist.get_global_rank(group, r)
                    ops.append(dist.P2POp(dist.irecv, land[r_lo:r_hi], src, group))
            for w in (dist.batch_isend_irecv(ops) if ops else []):
                w.wait()
            if staged:
                full[hi - lo:n] = land[hi - lo:n].to(full.device)
            return full[:n]
        if hi > lo:
            dst = 0 if group is None else dist.get_global_rank(group, 0)
            for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, wire.contiguous(), dst, group)]):
                w.wait()
        return mine
    raise ValueError("gather must be 'root', 'all' or 'none'")
