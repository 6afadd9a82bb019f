#!/bin/bash
# Runs ON THE GPU BOX: SQ / LDS counter passes of the headline workload (steady launches only), summary to stdout.
TAG=${1:-r03a}; OUT=gpurun_out/prof_${TAG}_rmat10m_biased; mkdir -p "$OUT"; export TMPDIR=/tmp
BENCH="python3 bench.py --workload rmat10m_biased --steps 3 --warmup 2 --no-cpu-baseline --steady"
for PASS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC"; do
  NAME=$(echo $PASS | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --pmc $PASS --output-format csv -d "$OUT/pmc_$NAME" -o pmc -- $BENCH > /dev/null 2>> "$OUT/log.txt" || echo "pmc pass $NAME rc=$?" >> "$OUT/log.txt"
done
python3 scripts/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
grep "biased_pool" "$OUT/summary.txt" | cut -c1-110
tail -3 "$OUT/log.txt"
