#!/usr/bin/env python3
"""Times the triples walk of one libtrw_hip.so build (GPU box, tuning aid; cf. tune_elist.py): the bench's synthetic
knowledge graph (5 M triples / 1 M entities / 1000 relations), one walker per entity, L = 80."""
import argparse
import ctypes
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=os.path.join(ROOT, "torch_random_walk_amd", "libtrw_hip.so"))
ap.add_argument("--entities", type=int, default=1_000_000)
ap.add_argument("--triples", type=int, default=5_000_000)
ap.add_argument("--relations", type=int, default=1000)
ap.add_argument("--runs", type=int, default=5)
ap.add_argument("--L", type=int, default=80)
ap.add_argument("--walkers-mult", type=int, default=1)
ap.add_argument("--tag", default="")
args = ap.parse_args()
import torch  # noqa: E402


def load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "torch_random_walk_amd", name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


graphs, utils = load("graphs"), load("utils")
L = ctypes.CDLL(args.lib)
vp, i64, i32, dbl = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_double
L.trw_last_error.restype = ctypes.c_char_p
L.trw_walk_triples_workspace_bytes.argtypes = [i64, i64, i64, i32, i64, i32]
L.trw_walk_triples_workspace_bytes.restype = ctypes.c_size_t
L.trw_walk_triples_hip.argtypes = [vp, i64, vp, i64, vp, i64, i64, i32, i64, i32, i32, vp, vp, ctypes.c_size_t, vp]
L.trw_profile_enable.argtypes = [ctypes.c_int]
L.trw_profile_last.argtypes = [ctypes.POINTER(dbl)] * 3
dev = torch.device("cuda", 0)
n = args.entities
triples, entities, pad = graphs.synthetic_kg(n, args.triples, args.relations, seed=1, device=dev)
rt_index, triples = utils.build_relation_tail_index(triples, entities)
T = int(triples.shape[0])
tg = torch.arange(n, device=dev).repeat(args.walkers_mult)
nw = int(tg.numel())
stream = torch.cuda.current_stream().cuda_stream
nb = L.trw_walk_triples_workspace_bytes(T, n, nw, args.L, pad, 2)
ws = torch.empty(nb, dtype=torch.uint8, device=dev)
out = torch.empty((nw, 2 * args.L + 1), dtype=torch.int64, device=dev)
L.trw_profile_enable(1)
pre, wk, chk = [], [], None
for r in range(args.runs):
    rc = L.trw_walk_triples_hip(triples.data_ptr(), T, rt_index.data_ptr(), n, tg.data_ptr(), nw, 0, args.L, pad, 10, 2,
                                out.data_ptr(), ws.data_ptr(), nb, stream)
    assert rc == 0, L.trw_last_error()
    a, b, c = dbl(), dbl(), dbl()
    assert L.trw_profile_last(ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)) == 0
    pre.append(a.value)
    wk.append(b.value)
    s = int(out.sum())
    chk = s if chk is None else chk
    assert chk == s
live = int((out[:, 1::2] != pad).sum())
print("%-40s shadow %.3f  walk %s  best %.3f ms  live hops %.3e of %.3e  checksum %x" % (
    args.tag or os.path.basename(args.lib), min(pre), " ".join("%.3f" % t for t in wk), min(wk), live, nw * args.L,
    chk & 0xffffffff), flush=True)
