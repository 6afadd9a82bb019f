#!/bin/bash
# GPU box: per-kernel times of one prepare level (scripts/tune_biased.py --what W) under rocprofv3 --kernel-trace --stats.
# usage: scripts/prof_what.sh W
O=$GRAFT_REPO_ROOT/gpurun_out/prof_what_$1; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o p -- python3 $GRAFT_REPO_ROOT/scripts/tune_biased.py --what $1 --runs 2 > $O/run.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - $1 <<'PY'
import csv, glob, sys
for f in glob.glob("gpurun_out/prof_what_%s/**/*kernel_stats.csv" % sys.argv[1], recursive=True):
    for r in list(csv.DictReader(open(f)))[:40]:
        if "at::" in r["Name"] or "rocprim" in r["Name"] or "rocclr" in r["Name"]: continue
        print("%-72s calls %3s avg %10.3f ms" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
rm -f $O/*trace.csv
