#!/bin/bash
set -o pipefail
O=gpurun_out/prep2; rm -rf $O; mkdir -p $O
for sl in 2 3; do
cd /tmp && export TMPDIR=/tmp
TRW_HASH_SLACK=$sl rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof$sl -o t -- python3 $GRAFT_REPO_ROOT/scripts/tune_biased.py --what 14 --runs 2 > $GRAFT_REPO_ROOT/$O/prof$sl.log 2>&1
cd $GRAFT_REPO_ROOT
echo "== slack $sl"
python3 - $sl <<'PY'
import csv, glob, sys
for f in glob.glob("gpurun_out/prep2/prof%s/**/*kernel_stats.csv" % sys.argv[1], recursive=True):
    for r in list(csv.DictReader(open(f)))[:18]:
        if "at::" in r["Name"] or "rocprim" in r["Name"]: continue
        print("%-60s calls %4s avg %10.3f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
done
rm -rf $O/prof*/
