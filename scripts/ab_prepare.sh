#!/bin/bash
# Runs ON THE GPU BOX: the prepare pass of R-MAT 10M/110M under different settings (VARIANTS: space-separated
# "K=V,K=V" groups), the walk over each (same checksum = same masks, as far as the walk can tell), and the per-kernel
# times of the default one.
set -o pipefail
O=gpurun_out/prep; mkdir -p $O; rm -f $O/ab.txt
timeout -k 10 600 python -m pytest tests/test_gpu_prepared.py -x -q -k "tile_pass or c_abi or dropped or unsorted_rows_with_marks" > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
for what in ${WHATS:-6 14}; do
  for v in ${VARIANTS:-TRW_COMMON_MAX=8}; do
    timeout -k 10 300 python scripts/tune_biased.py --what $what --runs ${RUNS:-3} --env ${v//,/ } --tag "what=$what $v" 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt || exit 1
  done
done
[ -n "$NOPROF" ] && exit 0
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o tile -- python3 $GRAFT_REPO_ROOT/scripts/tune_biased.py --what 14 --runs 2 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/prep/prof/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:16]:
        if "at::" in r["Name"] or "rocprim" in r["Name"]: continue
        print("%-60s calls %4s avg %10.3f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
rm -f $O/prof/*trace.csv; du -sh $O
