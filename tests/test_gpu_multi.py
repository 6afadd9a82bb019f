"""The N > 1 path on the HIP kernels (pytest -m gpu): shards through `walk_sharded` with two real ranks,
bench.py's own launcher, and BASELINE configs[3] (111 M nodes / 1.6 B edges) walked shard by shard."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


_WORKER = r"""
import os, sys, torch
import torch.distributed as dist
sys.path.insert(0, {root!r})
rank, world, port, gather, p, q, outdir = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], \
    float(sys.argv[5]), float(sys.argv[6]), sys.argv[7]
wire, chunks = sys.argv[8], int(sys.argv[9])
narrow_result = wire == "narrow"
from torch_random_walk_amd import graphs
from torch_random_walk_amd.dist import walk_sharded
torch.cuda.set_device(0)                      # one card on this box: both ranks share it
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
rp, ci = graphs.rmat(3000, 30000, seed=4, device="cuda:0")
tg = torch.arange(3000, device="cuda:0").repeat(7)[:20001]        # not divisible by the world size
res = walk_sharded(rp, ci, tg, p, q, 33, 42, gather=gather, wire="auto" if narrow_result else wire, chunks=chunks,
                   result_dtype=torch.int32 if narrow_result else torch.int64)   # walk_fn=None: the HIP path
assert res.is_cuda and res.dtype == (torch.int32 if narrow_result else torch.int64)
if narrow_result:          # narrow rows end to end (written by the walk kernels, landed in the result): the same ids
    from torch_rw import rw
    res = rw.widen_rows(res.contiguous())
torch.save(res.cpu(), os.path.join(outdir, "r%d.pt" % rank))
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.parametrize("gather,wire,chunks", [("root", "auto", 1), ("all", "auto", 1), ("root", "int64", 1),
                                                ("root", "uint32", 4), ("all", "int64", 1),
                                                ("root", "narrow", 1), ("root", "narrow", 4), ("all", "narrow", 1)])
@pytest.mark.parametrize("p,q", [(1.0, 1.0), (0.25, 4.0)])
def test_sharded_walk_two_ranks_on_the_hip_path(tmp_path, gather, wire, chunks, p, q):
    """world_size 2, both ranks on cuda:0 with the HIP kernels doing the walking (the gloo transport
    stages the blocks through the host; under RCCL the same code hands device tensors over): the assembled
    result equals the single call and the oracle -- whether the blocks travel as int64 or in the uint32 wire
    format (narrowed and widened by the library's kernels), in one piece or in four row chunks -- or, "narrow"
    (result_dtype=torch.int32), as narrow rows end to end: written by the walk kernels, landed in an int32 result."""
    from torch_random_walk_amd import graphs
    from torch_rw import rw
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", str(port), gather, str(p), str(q),
                               str(tmp_path), wire, str(chunks)], env=env) for r in range(2)]
    codes = [pr.wait(timeout=600) for pr in procs]
    assert codes == [0, 0]
    rp, ci = graphs.rmat(3000, 30000, seed=4, device=DEV)
    tg = torch.arange(3000, device=DEV).repeat(7)[:20001]
    mode = rw.RNG_GLIBC if (p == 1.0 and q == 1.0) else rw.RNG_WALKER
    single = rw.walk_ex(rp, ci, tg, p, q, 33, 42, rng_mode=mode).cpu()
    omode = O.RNG_GLIBC_SEQ if (p == 1.0 and q == 1.0) else O.RNG_WALKER
    ref = torch.from_numpy(O.walk(rp.cpu().numpy(), ci.cpu().numpy(), tg.cpu().numpy(), p, q, 33, 42, rng_mode=omode))
    assert torch.equal(single, ref)
    r0 = torch.load(os.path.join(tmp_path, "r0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "r1.pt"))
    assert torch.equal(r0, single)
    if gather == "all":
        assert torch.equal(r1, single)
    else:
        assert torch.equal(r1, single[10001:])


_RCCL_SMOKE = r"""
import os, sys, torch
import torch.distributed as dist
sys.path.insert(0, {root!r})
port = int(sys.argv[1])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)          # bench.py's init call
x = torch.arange(1 << 20, device=dev, dtype=torch.int64)
dist.all_reduce(x, op=dist.ReduceOp.MAX)                                      # the replica checksum's collective
assert int(x[-1]) == (1 << 20) - 1
dist.barrier()
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t) == 1.5
# the gather's transport: a group of point-to-point transfers of DEVICE tensors in the int32 wire dtype (to itself:
# the one peer a single card has)
from torch_rw import rw
rows = torch.randint(0, 2**32, (4096, 81), device=dev, dtype=torch.int64)
wire = rw.narrow_rows(rows)
land = torch.empty_like(wire)
works = dist.batch_isend_irecv([dist.P2POp(dist.irecv, land, 0), dist.P2POp(dist.isend, wire, 0)])
for w in works:
    w.wait()
assert torch.equal(rw.widen_rows(land), rows)
got = torch.empty((1 * 4096, 81), dtype=torch.int32, device=dev)
dist.all_gather_into_tensor(got, wire)
assert torch.equal(got, wire)
torch.cuda.synchronize()
dist.destroy_process_group()
print("rccl smoke ok")
"""


def test_rccl_smoke_world_1(tmp_path):
    """librccl loads and the calls the N > 1 path makes work on this box, in a fresh child process: init_process_group
    ("nccl", device_id=...) as bench.py does it, device all_reduce, barrier, a batch_isend_irecv group of device tensors
    in the wire dtype, all_gather_into_tensor, destroy.  One card, so world_size 1."""
    script = tmp_path / "rccl_smoke.py"
    script.write_text(_RCCL_SMOKE.format(root=ROOT))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, str(script), str(_free_port())], env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert "rccl smoke ok" in out.stdout


def test_walk_sharded_world_1_rccl(tmp_path):
    """walk_sharded itself under backend "nccl" (world_size 1: the code path of a rank that is both root and only
    member -- init, the walk on the HIP path, destroy)."""
    script = tmp_path / "w1.py"
    script.write_text(r"""
import os, sys, torch
import torch.distributed as dist
sys.path.insert(0, %r)
os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", sys.argv[1]
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from torch_random_walk_amd import graphs
from torch_random_walk_amd.dist import walk_sharded
from torch_rw import rw
rp, ci = graphs.rmat(3000, 30000, seed=4, device="cuda:0")
tg = torch.arange(3000, device="cuda:0")
for gather in ("root", "all", "none"):
    a = walk_sharded(rp, ci, tg, 0.25, 4.0, 20, 7, gather=gather, chunks=3)
    assert torch.equal(a, rw.walk_ex(rp, ci, tg, 0.25, 4.0, 20, 7, rng_mode=rw.RNG_WALKER))
dist.barrier()
dist.destroy_process_group()
print("ok")
""" % ROOT)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, str(script), str(_free_port())], env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` from a plain shell (no torchrun around it): the parent starts the ranks
    before it has touched the GPU and relays the one JSON line; gloo here because the box has one card."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
                          "--workload", "rmat1m_biased", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["value"] > 0
    assert line["config"]["global_walkers"] == 2 * line["config"]["walkers_per_gpu"]
    # the N > 1 line says what the gather cost and what travelled
    mg = line["multi_gpu"]
    assert mg["wire_dtype"] == "uint32" and mg["gather_chunks"] >= 1
    assert mg["bytes_per_rank"] == line["config"]["walkers_per_gpu"] * 81 * 4
    assert mg["gather_ms"] > 0 and mg["walk_kernel_ms"] > 0


def test_bench_line_carries_the_contract_fields():
    """One JSON line with the driver's fields, the roofline of the dominant kernel and the CPU baseline."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "rmat1m_biased", "--steps", "2",
                          "--warmup", "1", "--cpu-seconds", "1"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["metric"] == "walk-steps/s" and line["n_gpus"] == 1 and line["steps"] == 2 and line["warmup"] == 1
    assert line["higher_is_better"] is True and line["scaling"] == "weak" and line["vs_baseline"] is None
    assert line["dtype"] == "int64" and "synthetic" in line["data"] and line["config"]["workload"] == "rmat1m_biased"
    roof = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9 and roof["kernel"] == "biased_pool_kernel"
    # what the prepared graph costs a caller who does not walk it for ever
    am = roof["amortised"]
    assert am["value_one_shot_call"] > 0 and am["value_incl_prepare_at_10_epochs"] > 0
    assert am["value_incl_prepare_at_10_epochs"] < line["value"] and "epochs_to_break_even" in am
    cpu = line["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cpu, key
    assert cpu["kind"] == "port" and cpu["value"] > 0 and cpu["unit"] == line["unit"]
    assert line["value"] > 1e3 * cpu["value"]


def _sparse_host_csr(rp, ci, nodes):
    """Host CSR with the same node count in which only `nodes` keep their rows (the others are empty, so an
    oracle walk that left the GPU's path would stop with a degree-0 error): lets the oracle check rows of a
    14 GB graph without moving the graph."""
    n = rp.numel() - 1
    nodes_d = torch.as_tensor(nodes, device=rp.device)
    starts, ends = rp[nodes_d], rp[nodes_d + 1]
    deg = torch.zeros(n, dtype=torch.int64, device=rp.device)
    deg[nodes_d] = ends - starts
    rp_s = torch.zeros(n + 1, dtype=torch.int64, device=rp.device)
    torch.cumsum(deg, 0, out=rp_s[1:])
    order = torch.argsort(nodes_d).tolist()
    starts_h, ends_h = starts.tolist(), ends.tolist()
    pieces = [ci[starts_h[k]:ends_h[k]] for k in order]
    ci_s = torch.cat(pieces) if pieces else torch.empty(0, dtype=torch.int64, device=rp.device)
    return rp_s.cpu().numpy(), ci_s.cpu().numpy()


def test_config4_papers100m_sharded_equals_unsharded():
    """BASELINE configs[3]: ogbn-papers100M-scale synthetic CSR (111 M nodes / 1.6 B edges + ring), p=q=1,
    L=80, every node a start node, the batch split into the 8 walker_base shards the 8-GPU job uses --
    walked here one shard after another on the single card, each straight into its slice of the gathered
    result (`walk_ex(out=...)`).  The assembled blocks equal ONE unsharded call (a different kernel
    family: the big call walks fat entries, a shard the compact shadow), every step of every walk is an
    edge, and sampled rows (first, last, both sides of every shard boundary) equal the oracle's
    reference-exact glibc stream."""
    from torch_random_walk_amd import graphs
    from torch_random_walk_amd.dist import shard_bounds
    from torch_rw import rw
    n, L, seed, world = 111_000_000, 80, 10, 8
    rp, ci = graphs.rmat(n, 1_600_000_000, seed=1, device=DEV)
    assert int(rp[-1]) == ci.numel() == 1_600_000_000 + n
    tg = torch.arange(n, device=DEV)
    full = rw.walk_ex(rp, ci, tg, 1.0, 1.0, L, seed, rng_mode=rw.RNG_GLIBC)        # 72 GB
    assert tuple(full.shape) == (n, L + 1)
    torch.cuda.empty_cache()

    # -- shard blocks == slices of the unsharded call ------------------------------------------------
    keys = torch.repeat_interleave(torch.arange(n, device=DEV), rp[1:] - rp[:-1]).mul_(n).add_(ci)
    assert bool((keys[1:] >= keys[:-1]).all())           # rows sorted: the key list needs no sort
    bounds = [shard_bounds(n, world, r) for r in range(world)]
    assert bounds[0][0] == 0 and bounds[-1][1] == n
    for r, (lo, hi, block) in enumerate(bounds):
        blk = torch.empty((hi - lo, L + 1), dtype=torch.int64, device=DEV)
        got = rw.walk_ex(rp, ci, tg[lo:hi], 1.0, 1.0, L, seed, walker_base=lo, rng_mode=rw.RNG_GLIBC, out=blk)
        assert got.data_ptr() == blk.data_ptr()
        assert torch.equal(blk, full[lo:hi]), "shard %d differs from the unsharded call" % r
        assert torch.equal(blk[:, 0], tg[lo:hi])
        for s in range(L):                               # every step is an edge of the graph
            qk = blk[:, s] * n + blk[:, s + 1]
            pos = torch.searchsorted(keys, qk).clamp_(max=keys.numel() - 1)
            assert bool((keys[pos] == qk).all()), "shard %d column %d leaves the graph" % (r, s + 1)
        del blk, got
    del keys
    torch.cuda.empty_cache()

    # -- sampled rows against the oracle -------------------------------------------------------------
    rows = {0, 1, n - 2, n - 1}
    for lo, hi, _ in bounds:
        rows.update(x for x in (lo - 1, lo, lo + 1, hi - 1) if 0 <= x < n)
    rows.update(np.random.default_rng(0).integers(0, n, 24).tolist())
    rows = np.array(sorted(rows))
    assert len(rows) >= 48
    got = full[torch.as_tensor(rows, device=DEV)].cpu().numpy()
    rp_s, ci_s = _sparse_host_csr(rp, ci, np.unique(got))
    for r, g in zip(rows, got):
        ref = O.walk(rp_s, ci_s, np.array([r]), 1.0, 1.0, L, seed, rng_mode=O.RNG_GLIBC_INDEXED, walker_base=int(r))
        assert (ref[0] == g).all(), "row %d differs from the oracle" % r


def test_config4_graph_node2vec_shard_keeps_the_table_path():
    """The node2vec walk at BASELINE configs[3] scale: 1.71 G entries used to be beyond the edge
    table's 2^31-slot cap, which silently sent such graphs to the bisection kernel; then the table had 2^32 slots of
    16 bytes (69 GB).  Round 4: an edge set of 8-byte keys, 20 bytes per edge (34 GB) -- the prepared graph is 96 GB,
    56 bytes per edge, one GPU's shard of the start nodes (13.9 M walkers, global walker indices of shard 5) walks on
    the walker-pool kernel, and sampled rows equal the oracle's per-walker stream."""
    from torch_random_walk_amd import graphs
    from torch_random_walk_amd.dist import shard_bounds
    from torch_rw import rw
    rw.clear_cache()
    n, L, seed = 111_000_000, 80, 10
    rp, ci = graphs.rmat(n, 1_600_000_000, seed=1, device=DEV)
    lo, hi, _ = shard_bounds(n, 8, 5)
    tg = torch.arange(lo, hi, device=DEV)
    w = rw.walk_ex(rp, ci, tg, 0.25, 4.0, L, seed, walker_base=lo, rng_mode=rw.RNG_WALKER, check=True)
    info = rw.cache_info()
    assert len(info) == 1 and info[0][2] & rw.PREP_BIASED, info
    assert 56 * ci.numel() < info[0][3] < 57 * ci.numel()        # shadow 4 + edge set 20 + entries 32 bytes per edge
    assert torch.equal(w[:, 0], tg) and int(w.min()) >= 0 and int(w.max()) < n
    rows = np.unique(np.concatenate([[0, 1, hi - lo - 1], np.random.default_rng(0).integers(0, hi - lo, 21)]))
    got = w[torch.as_tensor(rows, device=DEV)].cpu().numpy()
    rp_s, ci_s = _sparse_host_csr(rp, ci, np.unique(got))
    for r, g in zip(rows, got):
        ref = O.walk(rp_s, ci_s, np.array([lo + r]), 0.25, 4.0, L, seed, rng_mode=O.RNG_WALKER, walker_base=int(lo + r))
        assert (ref[0] == g).all(), "row %d differs from the oracle" % r
    del w
    rw.clear_cache()
    torch.cuda.empty_cache()


def test_wire_format_kernels_round_trip():
    """trw_rows_narrow_hip / trw_rows_widen_hip through the extension: any size (the last count % 4 values, blocks that
    start at an odd element of a larger tensor take the scalar kernels), the whole uint32 range, overflow only reported
    in strict mode; equal to the host legs."""
    from torch_rw import rw
    g = torch.Generator(device=DEV).manual_seed(5)
    for shape in [(0, 81), (1, 1), (3,), (1023, 81), (4096, 81), (100_003, 7)]:
        x = torch.randint(0, 2**32, shape, generator=g, dtype=torch.int64, device=DEV)
        if x.numel():
            x.view(-1)[-1] = 2**32 - 1
        w = rw.narrow_rows(x, strict=True)
        assert w.dtype == torch.int32 and w.shape == x.shape and torch.equal(rw.widen_rows(w), x)
        assert torch.equal(w.cpu(), rw.narrow_rows(x.cpu()))
    big = torch.arange(1000, dtype=torch.int64, device=DEV) * 7
    src, dst = big[3:], torch.empty(1000, dtype=torch.int32, device=DEV)[1:998]      # both unaligned windows
    assert torch.equal(rw.widen_rows(rw.narrow_rows(src, out=dst)), src)
    bad = torch.tensor([1, 2**32, 3, 4, 5], device=DEV)
    assert rw.narrow_rows(bad).tolist() == [1, 0, 3, 4, 5]
    with pytest.raises(RuntimeError, match="does not fit"):
        rw.narrow_rows(bad, strict=True)
