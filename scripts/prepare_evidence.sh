#!/bin/bash
# Runs ON THE GPU BOX: the evidence DESIGN.md's "tile pass" section quotes -- the prepare pass of R-MAT 10M/110M with the
# tile pass and with round 3's wave-per-entry pass, the sweep of the inline cap, the per-kernel times of the default,
# and the PMC passes of heavy_tile_kernel and csr_ext32_kernel.  Output: gpurun_out/collected/<tag>_prepare_ab.txt,
# <tag>_tile_pmc.txt
TAG=${1:-r04}
O=gpurun_out/collected; mkdir -p $O
{
echo "# scripts/tune_biased.py --what W --runs 2 --env ...: prepare = all kernels of trw_prepare_csr_hip (events), walk = biased_pool_kernel"
for what in 2 6 14; do
  for v in TRW_HEAVY_TILE=1 TRW_HEAVY_TILE=0; do
    [ $what = 2 ] && [ $v = TRW_HEAVY_TILE=0 ] && continue
    timeout -k 10 300 python scripts/tune_biased.py --what $what --runs 2 --env $v --tag "what=$what $v" 2>&1 | grep -v amdgpu.ids
  done
done
echo "# the inline cap (rows the owning thread marks itself), full marks"
for c in 4 8 16 32; do
  timeout -k 10 300 python scripts/tune_biased.py --what 14 --runs 2 --env TRW_COMMON_MAX=$c --tag "what=14 TRW_COMMON_MAX=$c" 2>&1 | grep -v amdgpu.ids
done
echo "# places of the edge set per edge, in tenths (bytes per edge of the whole prepared graph: 58 / 64 / 68)"
for s in 20 25 30; do
  timeout -k 10 300 python scripts/tune_biased.py --what 14 --runs 2 --env TRW_HASH_SLACK=$s --tag "what=14 TRW_HASH_SLACK=$s" 2>&1 | grep -v amdgpu.ids
done
} > $O/${TAG}_prepare_ab.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prep_prof -o t -- python3 $GRAFT_REPO_ROOT/scripts/tune_biased.py --what 14 --runs 2 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 - >> $O/${TAG}_prepare_ab.txt <<'PY'
import csv, glob
print("# rocprofv3 --kernel-trace --stats of scripts/tune_biased.py --what 14 --runs 2 (defaults): the prepare pass's kernels and the walk")
for f in glob.glob("gpurun_out/prep_prof/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:20]:
        if "at::" in r["Name"] or "rocprim" in r["Name"]: continue
        print("%-64s calls %3s avg %9.3f ms" % (r["Name"].replace("(anonymous namespace)::", "")[:64], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
rm -rf gpurun_out/prep_prof
{ echo "# PMC passes (rocprofv3 --pmc, one pass per counter group) over scripts/tune_biased.py --what 14 --runs 1: sums over the kernel's one dispatch"
  echo "## heavy_tile_kernel"; bash scripts/pmc_prepare.sh heavy_tile_kernel 2>/dev/null
  echo "## csr_ext32_kernel"; bash scripts/pmc_prepare.sh csr_ext32_kernel 2>/dev/null
} > $O/${TAG}_tile_pmc.txt
tail -3 $O/${TAG}_prepare_ab.txt
