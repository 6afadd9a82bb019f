#!/usr/bin/env python3
"""Times the biased CSR walk of one libtrw_hip.so build over a prepared graph (GPU box, tuning aid).

    python scripts/tune_biased.py [--lib PATH] [--nodes N --edges M] [--what 6] [--runs 4] [--env K=V ...]

Loads the library with ctypes only (no torch extension), so differently compiled variants can be timed
side by side: scripts/tune_variants.sh builds them with -D flags.  Prints one line per run."""
import argparse
import ctypes
import importlib.util
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=os.path.join(ROOT, "torch_random_walk_amd", "libtrw_hip.so"))
ap.add_argument("--nodes", type=int, default=10_000_000)
ap.add_argument("--edges", type=int, default=100_000_000)
ap.add_argument("--walkers", type=int, default=0)
ap.add_argument("--what", type=int, default=6)
ap.add_argument("--runs", type=int, default=4)
ap.add_argument("--L", type=int, default=80)
ap.add_argument("--p", type=float, default=0.25)
ap.add_argument("--q", type=float, default=4.0)
ap.add_argument("--env", nargs="*", default=[])
ap.add_argument("--tag", default="")
ap.add_argument("--count", action="store_true")
ap.add_argument("--narrow", action="store_true", help="uint32 rows straight from the walk kernel (trw_walk_csr_prepared_u32_hip)")
args = ap.parse_args()
for kv in args.env:
    k, v = kv.split("=", 1)
    os.environ[k] = v

import torch  # noqa: E402

spec = importlib.util.spec_from_file_location("graphs", os.path.join(ROOT, "torch_random_walk_amd", "graphs.py"))
graphs = importlib.util.module_from_spec(spec)
spec.loader.exec_module(graphs)

L = ctypes.CDLL(args.lib)
vp, i64, i32, dbl = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_double
L.trw_last_error.restype = ctypes.c_char_p
L.trw_prepare_csr_bytes.argtypes = [i64, i64, i32]
L.trw_prepare_csr_bytes.restype = ctypes.c_size_t
L.trw_prepare_csr_hip.argtypes = [vp, i64, vp, i64, i32, vp, ctypes.c_size_t, vp]
L.trw_walk_csr_prepared_workspace_bytes.argtypes = [i64, i32, dbl, dbl, i32]
L.trw_walk_csr_prepared_workspace_bytes.restype = ctypes.c_size_t
L.trw_walk_csr_prepared_hip.argtypes = [vp, ctypes.c_size_t, i32, vp, i64, vp, i64, vp, i64, i64, dbl, dbl, i32, i32,
                                        i32, vp, vp, ctypes.c_size_t, vp]
L.trw_walk_csr_prepared_u32_hip.argtypes = L.trw_walk_csr_prepared_hip.argtypes
L.trw_profile_enable.argtypes = [ctypes.c_int]
L.trw_profile_last.argtypes = [ctypes.POINTER(dbl)] * 3
L.trw_profile_prepare_last.argtypes = [ctypes.POINTER(dbl)]
L.trw_walk_counters.argtypes = [vp, vp, ctypes.POINTER(ctypes.c_uint64)]
L.trw_walk_status.argtypes = [vp, vp, ctypes.POINTER(i64)]

dev = torch.device("cuda", 0)
n = args.nodes
rp, ci = graphs.rmat(n, args.edges, seed=1, device=dev)
nnz = ci.numel()
B = args.walkers or n
tg = torch.arange(B, device=dev, dtype=torch.int64) % n
stream = torch.cuda.current_stream().cuda_stream
pb = L.trw_prepare_csr_bytes(n, nnz, args.what)
blob = torch.empty(pb, dtype=torch.uint8, device=dev)
L.trw_profile_enable(1)
rc = L.trw_prepare_csr_hip(rp.data_ptr(), n, ci.data_ptr(), nnz, args.what, blob.data_ptr(), pb, stream)
assert rc == 0, L.trw_last_error()
pm = dbl()
L.trw_profile_prepare_last(ctypes.byref(pm))
wb = L.trw_walk_csr_prepared_workspace_bytes(B, args.L, args.p, args.q, 2)
ws = torch.empty(wb, dtype=torch.uint8, device=dev)
out = torch.empty((B, args.L + 1), dtype=torch.int32 if args.narrow else torch.int64, device=dev)
walk_entry = L.trw_walk_csr_prepared_u32_hip if args.narrow else L.trw_walk_csr_prepared_hip
times = []
chk = None
for r in range(args.runs):
    if args.count and r == args.runs - 1:
        L.trw_profile_enable(3)
    rc = walk_entry(blob.data_ptr(), pb, args.what, rp.data_ptr(), n, ci.data_ptr(), nnz, tg.data_ptr(),
                                     B, 0, args.p, args.q, args.L, 10, 2, out.data_ptr(), ws.data_ptr(), wb, stream)
    assert rc == 0, L.trw_last_error()
    a, b, c = dbl(), dbl(), dbl()
    assert L.trw_profile_last(ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)) == 0
    times.append(b.value)
    s = int(out[:100000].long().sum()) ^ int(out[-100000:].long().sum())
    assert chk is None or chk == s, "runs differ"
    chk = s
st = (i64 * 4)()
L.trw_walk_status(ws.data_ptr(), stream, st)
extra = ""
if args.count:
    c = (ctypes.c_uint64 * 4)()
    L.trw_walk_counters(ws.data_ptr(), stream, c)
    extra = " counters/step: trials %.2f fetch %.3f tests %.3f probes %.3f" % tuple(x / (B * args.L) for x in c)
    raw = ws[:256].view(torch.int64).cpu().tolist()
    if st[3] == 5:  # the walker-pool kernel's wave counters (status[12..23])
        it, ap, al, ip, il, bp_, bl, tp, c0, c1, c2, c3 = raw[12:24]
        cyc = max(1, c0 + c1 + c2 + c3)
        extra += ("\n    pool: wave iterations %.3e; adopt passes/iter %.2f (%.1f lanes), issue %.2f (%.1f), burst %.2f (%.1f lanes, "
                  "%.2f pair passes, trial lane util %.1f%%); wall cycles adopt %.1f%% refill %.1f%% issue %.1f%% burst %.1f%% "
                  "(%.0f cycles / iteration)" % (it, ap / it, al / max(1, ap), ip / it, il / max(1, ip), bp_ / it, bl / max(1, bp_),
                                                 tp / max(1, bp_), 100.0 * c[0] / max(1, tp * 128), 100.0 * c0 / cyc, 100.0 * c1 / cyc,
                                                 100.0 * c2 / cyc, 100.0 * c3 / cyc, cyc / it))
        raw[12] = 0
    iters, reps, refills, ad = raw[12], raw[13], raw[14], raw[15]
    adopts, adopt_lanes = ad >> 40, ad & ((1 << 40) - 1)
    if iters:
        extra += ("\n    wave iterations %.3e, burst reps/iter %.2f (lane util %.1f%%), refill passes/iter %.2f, "
                  "iterations with adopt %.1f%% (lanes per adopt pass %.1f), trials per iteration per lane %.2f" % (
                      iters, reps / iters, 100.0 * c[0] / (reps * 64), refills / iters, 100.0 * adopts / iters,
                      adopt_lanes / max(1, adopts), c[0] / (iters * 64)))
print("%-28s prepare %.1f ms  walk ms %s  best %.2f  family %d status %d checksum %x%s" % (
    args.tag or os.path.basename(args.lib), pm.value, " ".join("%.2f" % t for t in times), min(times), st[3], st[0], chk & 0xffffffff,
    extra), flush=True)
