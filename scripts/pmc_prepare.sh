#!/bin/bash
# Runs ON THE GPU BOX: PMC passes over the prepare kernels (scripts/tune_biased.py --what 14 --runs 1), condensed.
# usage: scripts/pmc_prepare.sh [kernel-name-substring[;substring...]] [K=V env for the run ...]   (several kernels: one section each)
set -u
KEY=${1:-heavy_tile_kernel}; shift
O=gpurun_out/pmc_prep; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
CMD="python3 scripts/tune_biased.py --what 14 --runs 1"
for PASS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"; do
  NAME=$(echo $PASS | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --pmc $PASS --output-format csv -d "$O/pmc_$NAME" -o pmc -- $CMD > /dev/null 2>> "$O/log.txt" || echo "pmc pass $NAME rc=$?" >> "$O/log.txt"
done
python3 - "$O" "$KEY" <<'PY'
import csv, glob, sys, collections
out, keys = sys.argv[1], sys.argv[2].split(";")
rows = []
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
for key in keys:
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in rows:
        if key in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    if len(keys) > 1:
        print("## " + key)
    # one dispatch of the kernel per run: sums over the rows of that dispatch (rocprofv3 emits one row per counter)
    for k in sorted(acc):
        print("%-34s %.4e" % (k, acc[k][0]))
PY
rm -rf $O/pmc_*
