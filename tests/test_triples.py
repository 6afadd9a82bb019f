"""Triples (knowledge-graph) walk: oracle pinned on the reference's golden, host index builder,
and (-m gpu) HIP parity.  Reference: csrc/cpu/rw_cpu_triples.cpp, tests/test_rw_triples.py."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"


@pytest.fixture(scope="module")
def tgold():
    with open(os.path.join(ROOT, "tests", "golden", "reference_test_rw_triples.json")) as f:
        return json.load(f)


def _index_by_loop(triples_sorted, n):
    """(first, last) per head by the obvious loop -- what torch_rw/utils.py:91-120 computes."""
    idx = np.full((n, 2), -1, dtype=np.int64)
    for e, (h, _, _) in enumerate(triples_sorted.tolist()):
        if idx[h, 0] == -1:
            idx[h, 0] = e
        idx[h, 1] = e
    if len(triples_sorted) == 1:
        idx[triples_sorted[0][0], 1] = -1     # the reference's loop never closes a lone triple
    return idx


def _kg(n_entities, n_triples, n_rel, seed):
    from torch_random_walk_amd import graphs
    from torch_rw import utils
    triples, entities, pad = graphs.synthetic_kg(n_entities, n_triples, n_rel, seed=seed)
    index, triples_sorted = utils.build_relation_tail_index(triples, entities)
    return triples_sorted, index, entities, pad


# ---- CPU tier ------------------------------------------------------------------------------

def test_index_builder_matches_reference_fixture(tgold):
    from torch_rw import utils
    tt = torch.Tensor(tgold["triples"]["list"])           # float tensor, as the reference's test builds it
    ents = torch.Tensor(sorted(set(tt[:, 0].tolist() + tt[:, 2].tolist()))).to(int)
    index, triples_sorted = utils.build_relation_tail_index(tt, ents)
    assert index.dtype == torch.int64 and triples_sorted.dtype == torch.int64
    assert index.tolist() == tgold["relation_tail_index"]["rows"]
    assert (triples_sorted[:, 0][1:] >= triples_sorted[:, 0][:-1]).all()


def test_index_builder_matches_loop_on_random_kg():
    ts, index, ents, pad = _kg(300, 2000, 7, 3)
    assert (index.numpy() == _index_by_loop(ts.numpy(), 300)).all()
    assert int((index[:, 0] == -1).sum()) > 0              # some entities head nothing
    from torch_rw import utils
    one = torch.tensor([[2, 9, 1]])
    idx1, _ = utils.build_relation_tail_index(one, torch.arange(4))
    assert idx1.tolist() == [[-1, -1], [-1, -1], [0, -1], [-1, -1]]


@pytest.mark.parametrize("mode", [O.RNG_GLIBC_SEQ, O.RNG_LIBC])
def test_oracle_reproduces_reference_golden(tgold, mode):
    from torch_rw import utils
    tt = torch.tensor(tgold["triples"]["list"])
    index, ts = utils.build_relation_tail_index(tt, torch.arange(5))
    c = tgold["walk_cpu"]
    w = O.walk_triples(ts.numpy(), index.numpy(), c["target_nodes"], c["walk_length"], c["padding_idx"], c["seed"],
                       rng_mode=mode)
    assert w.tolist() == c["walks"]


def test_oracle_draw_accounting_and_sharding():
    ts, index, ents, pad = _kg(200, 900, 5, 5)
    tg = np.random.default_rng(0).integers(0, 200, 500)
    a, da = O.walk_triples(ts.numpy(), index.numpy(), tg, 12, pad, 7, rng_mode=O.RNG_GLIBC_SEQ, return_draws=True)
    b, db = O.walk_triples(ts.numpy(), index.numpy(), tg, 12, pad, 7, rng_mode=O.RNG_LIBC, return_draws=True)
    assert (a == b).all() and da == db
    assert da < 500 * 12                                    # dead ends and single-edge hops draw nothing
    assert (a[:, 0] == tg).all() and a.shape == (500, 25)
    # padding is absorbing
    padded = a[:, 2::2] == pad
    assert (padded[:, 1:] >= padded[:, :-1]).all()
    full = O.walk_triples(ts.numpy(), index.numpy(), tg, 12, pad, 7, rng_mode=O.RNG_WALKER)
    parts = [O.walk_triples(ts.numpy(), index.numpy(), tg[lo:hi], 12, pad, 7, rng_mode=O.RNG_WALKER, walker_base=lo)
             for lo, hi in ((0, 7), (7, 300), (300, 500))]
    assert (np.concatenate(parts) == full).all()


def test_surface_names():
    from torch_rw import rw, utils
    import inspect
    import torch_rw_native
    assert list(inspect.signature(rw.walk_triples).parameters) == \
        ["triples_indexed", "relation_tail_index", "target_nodes", "walk_length", "padding_idx", "seed", "restart"]
    assert callable(torch_rw_native.walk_triples) and callable(utils.build_relation_tail_index)


def test_host_leg_reference_golden_and_oracle(tgold):
    """The reference's CPU golden (tests/test_rw_triples.py:70-79) through its own call on CPU tensors: the host leg
    (rw_init.cpp:66-74 -> triples::walk_triples_cpu), product code in csrc/trw_cpu.cpp; then against the oracle."""
    from torch_rw import rw, utils
    tt = torch.Tensor(tgold["triples"]["list"])
    ents = torch.Tensor(sorted(set(tt[:, 0].tolist() + tt[:, 2].tolist()))).to(int)
    index, ts = utils.build_relation_tail_index(tt, ents)
    c = tgold["walk_cpu"]
    w = rw.walk_triples(triples_indexed=ts, relation_tail_index=index, target_nodes=torch.tensor(c["target_nodes"]),
                        walk_length=c["walk_length"], seed=c["seed"], padding_idx=c["padding_idx"], restart=False)
    assert w.device.type == "cpu" and w.dtype == torch.int64 and w.tolist() == c["walks"]
    ts2, index2, ents2, pad = _kg(200, 900, 5, 5)
    tg = np.random.default_rng(0).integers(0, 200, 500)
    tg[::17] = pad
    ref = O.walk_triples(ts2.numpy(), index2.numpy(), tg, 12, pad, 7, rng_mode=O.RNG_GLIBC_SEQ)
    assert (rw.walk_triples(ts2, index2, torch.as_tensor(tg), 12, pad, 7).numpy() == ref).all()
    with pytest.raises(RuntimeError, match="outside"):
        rw.walk_triples(ts2, index2, torch.tensor([10**6]), 4, pad, 1)


# ---- GPU tier ------------------------------------------------------------------------------

@pytest.mark.gpu
def test_gpu_reference_golden(tgold):
    """The reference's own CPU golden through the reference's own call, on the HIP path."""
    from torch_rw import rw, utils
    tt = torch.Tensor(tgold["triples"]["list"])
    ents = torch.Tensor(sorted(set(tt[:, 0].tolist() + tt[:, 2].tolist()))).to(int)
    index, ts = utils.build_relation_tail_index(tt, ents)
    c = tgold["walk_cpu"]
    target_nodes = ents.repeat_interleave(2, 0)
    walks = rw.walk_triples(triples_indexed=ts.to(DEV), relation_tail_index=index.to(DEV),
                            target_nodes=target_nodes.to(DEV), walk_length=6, seed=10, padding_idx=c["padding_idx"],
                            restart=False)
    assert walks.dtype == torch.int64 and tuple(walks.shape) == (10, 13)
    assert walks.cpu().tolist() == c["walks"]               # reference tests/test_rw_triples.py:70-79


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{}, {"TRW_FORCE_WIDE": "1"}])
def test_gpu_walker_stream_equals_oracle(env, monkeypatch):
    from torch_rw import rw
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    ts, index, ents, pad = _kg(2000, 9000, 11, 9)
    tg = np.random.default_rng(1).integers(0, 2000, 5000)
    tg[::17] = pad                                           # start on the padding id
    for L in (1, 2, 5, 80):
        ref = O.walk_triples(ts.numpy(), index.numpy(), tg, L, pad, 3, rng_mode=O.RNG_WALKER, walker_base=1 << 33)
        got = rw.walk_triples_ex(ts.to(DEV), index.to(DEV), torch.as_tensor(tg).to(DEV), L, pad, 3,
                                 walker_base=1 << 33, rng_mode=rw.RNG_WALKER)
        assert (got.cpu().numpy() == ref).all()


@pytest.mark.gpu
def test_gpu_serial_replay_and_auto_rule():
    from torch_rw import rw
    ts, index, ents, pad = _kg(500, 2500, 6, 13)
    tg = np.random.default_rng(2).integers(0, 500, 3000)
    d = [ts.to(DEV), index.to(DEV), torch.as_tensor(tg).to(DEV)]
    ref = O.walk_triples(ts.numpy(), index.numpy(), tg, 80, pad, 10, rng_mode=O.RNG_GLIBC_SEQ)
    assert (rw.walk_triples_ex(*d, 80, pad, 10, rng_mode=rw.RNG_GLIBC).cpu().numpy() == ref).all()
    # AUTO: serial replay while N*L <= 2^18 (3000*80 = 240000), per-walker stream beyond
    assert (rw.walk_triples(*d, 80, pad, 10).cpu().numpy() == ref).all()
    big = rw.walk_triples(*d, 100, pad, 10).cpu().numpy()
    assert (big == O.walk_triples(ts.numpy(), index.numpy(), tg, 100, pad, 10, rng_mode=O.RNG_WALKER)).all()


@pytest.mark.gpu
def test_gpu_shards_errors_and_full_size():
    from torch_rw import rw, utils
    from torch_random_walk_amd import graphs
    ts, index, ents, pad = _kg(3000, 20000, 9, 17)
    d_ts, d_ix = ts.to(DEV), index.to(DEV)
    tg = torch.arange(3000, device=DEV).repeat(3)
    full = rw.walk_triples_ex(d_ts, d_ix, tg, 40, pad, 5, rng_mode=rw.RNG_WALKER)
    parts = [rw.walk_triples_ex(d_ts, d_ix, tg[lo:hi], 40, pad, 5, walker_base=lo, rng_mode=rw.RNG_WALKER)
             for lo, hi in ((0, 1), (1, 4097), (4097, 9000))]
    assert torch.equal(torch.cat(parts), full)
    assert tuple(rw.walk_triples(d_ts, d_ix, tg[:0], 4, pad, 1).shape) == (0, 9)
    with pytest.raises(RuntimeError, match="expected scalar type Long"):
        rw.walk_triples(d_ts.to(torch.int32), d_ix, tg, 4, pad, 1)
    with pytest.raises(RuntimeError, match="outside"):
        rw.walk_triples_ex(d_ts, d_ix, torch.tensor([3001], device=DEV), 4, pad, 1, check=True)   # not an entity, not padding
    rw.walk_triples(d_ts, d_ix, torch.tensor([3001], device=DEV), 4, pad, 1)          # default: reported later
    with pytest.raises(RuntimeError, match="outside"):
        rw.check_status()
    # BASELINE config 5 shape: 5M triples / 1M entities / 1000 relations, one walker per entity, L=80
    triples, ents5, pad5 = graphs.synthetic_kg(1_000_000, 5_000_000, 1000, seed=1, device=DEV)
    index5, ts5 = utils.build_relation_tail_index(triples, ents5)
    w1 = rw.walk_triples(ts5, index5, ents5, 80, pad5, 10)
    w2 = rw.walk_triples(ts5, index5, ents5, 80, pad5, 10)
    assert torch.equal(w1, w2) and torch.equal(w1[:, 0], ents5)
    # every hop is a triple of the graph (or padding after a dead end)
    n = 1_000_000
    key = (ts5[:, 0] * (n + 2000) + ts5[:, 1]) * n + ts5[:, 2]
    key, _ = torch.sort(key)
    for h in (0, 1, 40, 79):
        head, rel, tail = w1[:, 2 * h], w1[:, 2 * h + 1], w1[:, 2 * h + 2]
        live = rel != pad5
        q = (head[live] * (n + 2000) + rel[live]) * n + tail[live]
        pos = torch.searchsorted(key, q).clamp_(max=key.numel() - 1)
        assert bool((key[pos] == q).all())
        assert bool((tail[~live] == pad5).all())
    rows = np.unique(np.random.default_rng(0).integers(0, n, 300))
    ts_h, ix_h = ts5.cpu().numpy(), index5.cpu().numpy()
    got = w1[torch.as_tensor(rows, device=DEV)].cpu().numpy()
    for r, g in zip(rows, got):
        ref = O.walk_triples(ts_h, ix_h, [r], 80, pad5, 10, rng_mode=O.RNG_WALKER, walker_base=int(r))
        assert (ref[0] == g).all()
