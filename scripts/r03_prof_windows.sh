#!/bin/bash
export TMPDIR=/tmp; O=gpurun_out/r03pw; mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o stats -- python3 scripts/time_windows.py > $O/out.txt 2> $O/err.txt
tail -3 $O/out.txt
python3 - <<'PY'
import csv,glob
for p in glob.glob("gpurun_out/r03pw/stats/**/*kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(p)))
    for r in rows[:14]:
        import re
        m = re.search(r"(\w+kernel\w*(<[^(]*>)?)\(", r["Name"])
        print("%-60s calls %5s avg %10.1f us  tot %8.2f ms" % ((m.group(1) if m else r["Name"])[:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
