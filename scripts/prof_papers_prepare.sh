#!/bin/bash
# GPU box: per-kernel times of the full prepare of the BASELINE configs[3]-size graph (111 M nodes / 1.71 G entries) and of
# one shard's node2vec walk over it (rocprofv3 --kernel-trace --stats of scripts/tune_biased.py).
O=$GRAFT_REPO_ROOT/gpurun_out/papers_prep; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o p -- python3 $GRAFT_REPO_ROOT/scripts/tune_biased.py --nodes 111000000 --edges 1600000000 --walkers 13875000 --what 14 --runs 2 > $O/run.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/papers_prep/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:40]:
        if "at::" in r["Name"] or "rocprim" in r["Name"] or "rocclr" in r["Name"]: continue
        print("%-72s calls %3s avg %10.3f ms" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
grep -v amdgpu $O/run.log | tail -2
rm -f $O/*trace.csv
