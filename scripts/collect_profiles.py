#!/usr/bin/env python3
"""Copies the judged evidence out of gpurun_out/prof_<tag>_<workload>/ into profiles/:
the rocprofv3 --kernel-trace --stats CSV, the condensed summary, and the per-launch HBM traffic
of the dominant kernel (profiles/pmc_traffic.json, read by bench.py).

usage: scripts/collect_profiles.py <tag> <workload> [<workload> ...] [--dst DIR]
(--dst: write there instead of profiles/ -- on the GPU box, where only gpurun_out/ travels back)
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
argv = list(sys.argv)
DST = os.path.join(ROOT, "profiles")
if "--dst" in argv:
    k = argv.index("--dst")
    DST = os.path.abspath(argv[k + 1])
    del argv[k:k + 2]
tag = argv[1]
DOMINANT = {"elist_rmat1m_biased": ("elist_pool_kernel", "elist_walk_kernel"), "rmat10m_biased": ("biased_pool_kernel", "biased_mask_kernel", "biased_hash_kernel", "biased_walker_kernel"),
            "rmat10m_uniform": ("uniform_glibc_rows_kernel", "uniform_glibc_ext_kernel", "uniform_glibc_kernel"),
            "er100k_uniform": ("uniform_glibc_rows_kernel", "uniform_glibc_kernel", "uniform_glibc_ext_kernel"),
            "kg5m_triples": ("triples_walker_kernel",), 
            "elist_rmat1m_uniform": ("elist_walk_kernel",), "windows_1m": ("glibc_fill_mod_kernel",),
            "walk_windows_1m": ("uniform_glibc_rows_kernel",)}


def sources_sha():
    """sha256 over the kernel sources: bench.py quotes a PMC figure only while it was collected from these."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "torch_random_walk_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".hpp", ".cpp")) and name != "torch_rw_native.cpp":
            with open(os.path.join(d, name), "rb") as f:
                h.update(f.read())
    return h.hexdigest()


traffic_path = os.path.join(DST, "pmc_traffic.json")
base_path = traffic_path if os.path.exists(traffic_path) else os.path.join(ROOT, "profiles", "pmc_traffic.json")
traffic = json.load(open(base_path)) if os.path.exists(base_path) else {}
os.makedirs(DST, exist_ok=True)
for wl in argv[2:]:
    src = os.path.join(ROOT, "gpurun_out", "prof_%s_%s" % (tag, wl))
    dst = DST
    os.makedirs(dst, exist_ok=True)
    for name in ("stats/stats_kernel_stats.csv", "summary.txt"):
        p = os.path.join(src, name)
        if os.path.exists(p):
            shutil.copy(p, os.path.join(dst, "%s_%s_%s" % (tag, wl, os.path.basename(name))))
    # the steady-state dispatch of the dominant kernel in each PMC pass: the median of its launches, bench.py's small
    # check launches (under a tenth of the largest) left out.  The profiled command runs 2 warm-up + 3 timed steps over
    # the prepared graph, so they are the majority; the one cold first call (over a graph without its marks: 2.5x the
    # traffic) is an outlier the median ignores.  (Until r02 this kept only launches within a factor two of the LARGEST,
    # which picked the cold call once the steady state had become more than twice as lean.)
    vals = {}
    names = {}
    seen = defaultdict(lambda: defaultdict(list))   # kernel -> counter -> values of its dispatches
    for path in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                hit = [k for k in DOMINANT.get(wl, ()) if k + "<" in row["Kernel_Name"] or k + "(" in row["Kernel_Name"]]
                counting = hit and (hit[0].startswith("biased") or hit[0] == "elist_pool_kernel") and (hit[0] + "<true") in row["Kernel_Name"]
                if hit and not counting:   # (biased_*_kernel<true> is the counting variant, not the timed kernel)
                    seen[hit[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    # several candidates may have been launched (a kernel that finds the graph is not its kind returns at once):
    # the dominant one is the one that moved the most
    if seen:
        dom = max(seen, key=lambda k: max(seen[k].get("FETCH_SIZE", [0.0]) + seen[k].get("TCC_REQ_sum", [0.0])))
        for c, v in seen[dom].items():
            big = sorted(x for x in v if x >= 0.1 * max(v))
            vals[c] = big[len(big) // 2]
            names[c] = dom
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        # rocprofv3 reports both in KiB.  MI355X_MICROARCH.md (HBM): FETCH_SIZE = TCC_EA0_RDREQ x 64 B and
        # under-reports wide streaming reads by 2x; these kernels issue 4..16-byte random gathers, for which
        # one 64-byte request per L2 miss is what the fabric moves (5.6e10 misses/s x 128 B would exceed
        # the measured 6.3 TB/s), so no doubling is applied here ("Traffic").
        traffic[wl] = {
            "tag": tag,
            "sources_sha256": sources_sha(),
            "kernel": names.get("FETCH_SIZE", DOMINANT[wl][0]),
            "FETCH_SIZE_KiB": vals["FETCH_SIZE"], "WRITE_SIZE_KiB": vals["WRITE_SIZE"],
            "hbm_bytes_per_launch": (vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0,
            "TCC_REQ": vals.get("TCC_REQ_sum"), "TCC_HIT": vals.get("TCC_HIT_sum"), "TCC_MISS": vals.get("TCC_MISS_sum"),
            "TCP_TOTAL_CACHE_ACCESSES": vals.get("TCP_TOTAL_CACHE_ACCESSES_sum"),
        }
with open(traffic_path, "w") as f:
    json.dump(traffic, f, indent=1, sort_keys=True)
    f.write("\n")
print(json.dumps({k: traffic[k] for k in argv[2:] if k in traffic}, indent=1))
