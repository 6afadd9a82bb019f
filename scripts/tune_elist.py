#!/usr/bin/env python3
"""Times the biased edge-list walk of one libtrw_hip.so build (GPU box, tuning aid; cf. tune_biased.py)."""
import argparse
import ctypes
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=os.path.join(ROOT, "torch_random_walk_amd", "libtrw_hip.so"))
ap.add_argument("--nodes", type=int, default=1_000_000)
ap.add_argument("--edges", type=int, default=10_000_000)
ap.add_argument("--runs", type=int, default=4)
ap.add_argument("--L", type=int, default=80)
ap.add_argument("--env", nargs="*", default=[])
ap.add_argument("--tag", default="")
ap.add_argument("--count", action="store_true")
ap.add_argument("--walkers-mult", type=int, default=1, help="walkers per node (the drain of the pools at the end of a batch weighs less in a longer one)")
args = ap.parse_args()
for kv in args.env:
    k, v = kv.split("=", 1)
    os.environ[k] = v
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
L.trw_walk_edge_list_workspace_bytes.argtypes = [i64, i64, i64, i32, i64, i32]
L.trw_walk_edge_list_workspace_bytes.restype = ctypes.c_size_t
L.trw_walk_edge_list_hip.argtypes = [vp, i64, vp, i64, vp, i64, i64, dbl, dbl, i32, i32, i64, i32, i32, vp, vp, ctypes.c_size_t, vp]
L.trw_profile_enable.argtypes = [ctypes.c_int]
L.trw_profile_last.argtypes = [ctypes.POINTER(dbl)] * 3
L.trw_walk_status.argtypes = [vp, vp, ctypes.POINTER(i64)]
dev = torch.device("cuda", 0)
n = args.nodes
rp, ci = graphs.rmat(n, args.edges, seed=1, device=dev)
heads = torch.repeat_interleave(torch.arange(n, device=dev), rp[1:] - rp[:-1])
keep = ci != (heads + 1) % n
index, edges = utils.build_node_edge_index(torch.stack([heads[keep], ci[keep]], 1), torch.arange(n, device=dev))
E_ = edges.shape[0]
tg = torch.arange(n, device=dev).repeat(args.walkers_mult)
nw = int(tg.numel())
stream = torch.cuda.current_stream().cuda_stream
nb = L.trw_walk_edge_list_workspace_bytes(E_, n, nw, args.L, n, 2)
ws = torch.empty(nb, dtype=torch.uint8, device=dev)
out = torch.empty((nw, args.L + 1), dtype=torch.int64, device=dev)
L.trw_profile_enable(1)
pre, wk = [], []
chk = None
for r in range(args.runs):
    if args.count and r == args.runs - 1:
        L.trw_profile_enable(3)
    rc = L.trw_walk_edge_list_hip(edges.data_ptr(), E_, index.data_ptr(), n, tg.data_ptr(), nw, 0, 0.25, 4.0, args.L, 10, n, 1, 2,
                                  out.data_ptr(), ws.data_ptr(), nb, stream)
    assert rc == 0, L.trw_last_error()
    a, b, c = dbl(), dbl(), dbl()
    assert L.trw_profile_last(ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)) == 0
    pre.append(a.value)
    wk.append(b.value)
    s = int(out.sum())
    assert chk is None or chk == s
    chk = s
st = (i64 * 4)()
L.trw_walk_status(ws.data_ptr(), stream, st)
if args.count:
    raw = ws[:256].view(torch.int64).cpu().tolist()
    steps = nw * args.L
    names = ["RANGE", "ENTRY", "TEST", "MEMBER", "FIXNB", "PEEK", "NOENTRY"]
    print("   per step: trials %.2f fetch %.3f tests %.3f probes %.3f | gathers consumed by kind: %s | wave iterations %.3e" % (
        raw[4] / steps, raw[5] / steps, raw[6] / steps, raw[7] / steps,
        " ".join("%s %.3f" % (nm, raw[12 + k] / steps) for k, nm in enumerate(names)), raw[19]))
    cyc = max(1, sum(raw[20:24]))
    print("   phases: wall cycles adopt %.1f%% refill %.1f%% issue %.1f%% burst %.1f%% (%.0f cycles / iteration); burst passes/iter %.2f, "
          "%.1f lanes in, %.2f trial repetitions per burst at %.1f lanes; adopt waited for its gathers %.1f%% of the wall cycles" % (
              100.0 * raw[20] / cyc, 100.0 * raw[21] / cyc, 100.0 * raw[22] / cyc, 100.0 * raw[23] / cyc, cyc / max(1, raw[19]),
              raw[24] / max(1, raw[19]), raw[25] / max(1, raw[24]), raw[26] / max(1, raw[24]), raw[27] / max(1, raw[26]), 100.0 * raw[28] / cyc))
print("%-40s prepare %.2f  walk %s  best %.2f  family %d status %d checksum %x" % (
    args.tag or os.path.basename(args.lib), min(pre), " ".join("%.2f" % t for t in wk), min(wk), st[3], st[0], chk & 0xffffffff), flush=True)
