#!/bin/bash
# GPU box: the evidence DESIGN.md 4.1 quotes for the prepare pass -- per-level prepare / walk times and the per-kernel table
# (r05_prepare_ab.txt), the tile kernels' phase clocks and search statistics (r05_tile_phases.txt; needs variants/clk1.so and
# clk2.so from scripts/build_variant.sh), and their PMC counters (r05_tile_pmc.txt).  Results under gpurun_out/evidence/.
O=gpurun_out/evidence; mkdir -p $O
{
  echo "# scripts/tune_biased.py --what W --runs 2 (R-MAT 10M/110M): prepare = all kernels of trw_prepare_csr_hip (events), walk = biased_pool_kernel over that graph"
  for w in 2 6 14; do python3 scripts/tune_biased.py --what $w --runs 2 --tag "what=$w" 2>&1 | grep -v amdgpu; done
  echo "# the same with 3 M walkers (2.2 steps per entry)"
  for w in 2 6 14; do python3 scripts/tune_biased.py --what $w --runs 2 --walkers 3000000 --tag "what=$w 3M walkers" 2>&1 | grep -v amdgpu; done
  echo "# rocprofv3 --kernel-trace --stats of scripts/tune_biased.py --what 14 --runs 2: the prepare pass's kernels and the walk"
} > $O/r05_prepare_ab.txt
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o tile -- python3 $GRAFT_REPO_ROOT/scripts/tune_biased.py --what 14 --runs 2 > /dev/null 2>&1 )
python3 - >> $O/r05_prepare_ab.txt <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/evidence/prof/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:24]:
        if "at::" in r["Name"] or "rocprim" in r["Name"] or "rocclr" in r["Name"]: continue
        print("%-64s calls %3s avg %9.3f ms" % (r["Name"].replace("(anonymous namespace)::", "")[:64], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
rm -rf $O/prof
{
  echo "# -DTRW_TILE_CLOCKS=1 (the eight-wave kernel of the long tile rows) and =2 (the one-wave kernel of the short ones): wave cycles by phase, searches"
  for v in clk1 clk2; do python3 scripts/tune_biased.py --lib variants/$v.so --what 14 --runs 1 --tag "[$v]" 2>&1 | grep -v amdgpu; done
} > $O/r05_tile_phases.txt
{
  echo "# PMC passes (rocprofv3 --pmc, one pass per counter group) over scripts/tune_biased.py --what 14 --runs 1: sums over the kernel's one dispatch"
  bash scripts/pmc_prepare.sh "tile_bloom_kernel<2048;tile_bloom_kernel<256;entries_kernel;edge_table_build_kernel;marks_scatter_kernel" 2>/dev/null
} > $O/r05_tile_pmc.txt
tail -3 $O/r05_prepare_ab.txt; tail -2 $O/r05_tile_phases.txt; wc -l $O/r05_tile_pmc.txt
