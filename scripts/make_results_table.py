#!/usr/bin/env python3
"""Prints DESIGN.md's measurement table from the committed bench lines (profiles/<tag>_bench_*.json) and
profiles/pmc_traffic.json, so that the table is never typed by hand.  usage: scripts/make_results_table.py [tag]"""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
ORDER = ["default", "rmat10m_uniform", "er100k_uniform", "papers100m_uniform", "papers100m_biased", "papers100m_full", "kg5m_triples", "elist_rmat1m_uniform",
         "elist_rmat1m_biased", "windows_1m", "walk_windows_1m"]
pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
print("| workload | step ms (prepare / walk / finish) | units/s | dominant kernel | algorithmic GB/s (of 8 000) | "
      "HBM traffic per launch (PMC) | L2 misses/s vs gather ceiling | CPU baseline, 1 core → ratio |")
print("|---|---|---|---|---|---|---|---|")
for name in ORDER:
    path = os.path.join(ROOT, "profiles", "%s_bench_%s.json" % (tag, name))
    if not os.path.exists(path):
        continue
    d = json.load(open(path))
    r, c = d["roofline"], d.get("cpu_baseline") or {}
    wl = d["config"]["workload"]
    t = pmc.get(wl, {})
    traffic = "%.1f GB" % (t["hbm_bytes_per_launch"] / 1e9) if t else "—"
    rr = r.get("request_rate") or {}
    req = "%.2e (%.0f %%)" % (rr["achieved_per_s"], 100 * rr["frac"]) if rr else "—"
    cpu = "%.2e → %.0f×" % (c["value"], d["value"] / c["value"]) if c.get("value") else "—"
    print("| %s%s | %.2f (%.2f / %.2f / %.2f) | %.2e | `%s` | %.0f (%.1f %%) | %s | %s | %s |" % (
        wl, " (default)" if name == "default" else "", d["ms_per_step"], r["prepare_ms"], r["kernel_ms"], r["finish_ms"],
        d["value"], r["kernel"], r["achieved"], 100 * r["frac"], traffic, req, cpu))
