#!/usr/bin/env python3
"""Condenses a scripts/profile_gpu.sh output directory into a small text summary:
per-kernel time statistics from the kernel trace and per-kernel averages of every PMC counter."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    for k in ("uniform_glibc_rows_kernel", "uniform_glibc_ext_kernel", "uniform_glibc_kernel", "uniform_walker_kernel", "biased_walker_kernel", "glibc_windows_kernel",
              "csr_rows_sorted_kernel", "biased_glibc_seq_kernel", "transpose_rows_kernel", "csr_shadow_kernel",
              "tile_bloom_kernel", "entries_kernel", "marks_plan_kernel", "marks_scatter_kernel", "biased_hash_kernel", "biased_mask_kernel", "biased_pool_kernel", "edge_table_build_kernel",
              "csr_ext_kernel", "fill_kernel", "status_seed_kernel",
              "triples_windows_gather_kernel", "cbow_negative", "triples_glibc_seq_kernel", "elist_glibc_seq_kernel",
              "triples_walker_kernel", "triples_shadow_kernel", "elist_pool_kernel", "elist_ext_kernel", "elist_walk_kernel", "elist_shadow_kernel",
              "elist_table_build_kernel", "glibc_fill_mod_kernel", "windows_gather_kernel"):
        if k in name:
            tail = name[name.index(k):]
            return tail.split("(")[0][:44]
    return name.split("(")[0][-44:]


print("# kernel trace (all dispatches of the run, warm-up included)")
for path in glob.glob(os.path.join(out, "stats", "**", "*kernel_trace.csv"), recursive=True):
    durs = defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            durs[short(row["Kernel_Name"])].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
    tot = sum(sum(v) for v in durs.values())
    print("%-44s %6s %12s %12s %12s %7s" % ("kernel", "calls", "avg_ms", "min_ms", "max_ms", "pct"))
    for k, v in sorted(durs.items(), key=lambda kv: -sum(kv[1])):
        if sum(v) / tot < 0.002:
            continue
        print("%-44s %6d %12.4f %12.4f %12.4f %6.1f%%" % (k, len(v), sum(v) / len(v), min(v), max(v), 100 * sum(v) / tot))
for path in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    print("# rocprofv3 --stats file:", os.path.relpath(path, out))
    with open(path) as f:
        for i, line in enumerate(f):
            if i < 8:
                print("   ", line.rstrip()[:200])

print("\n# PMC counters: average per dispatch, per kernel (walk kernels only)")
for path in sorted(glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    acc = defaultdict(lambda: defaultdict(list))
    with open(path) as f:
        for row in csv.DictReader(f):
            k = short(row["Kernel_Name"])
            if not any(t in k for t in ("walk", "uniform", "transpose", "csr_", "glibc", "hash", "mask", "pool", "edge_table", "widen", "triples", "elist", "windows")):
                continue
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in acc.items():
        for c, v in sorted(cs.items()):
            print("%-40s %-32s n=%-3d avg=%.6g" % (k, c, len(v), sum(v) / len(v)))
