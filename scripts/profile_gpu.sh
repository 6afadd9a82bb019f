#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): rocprofv3 kernel-trace stats + separate PMC passes
# for one bench workload.  Usage: scripts/profile_gpu.sh <workload> <tag>
# Outputs land in gpurun_out/prof_<tag>/ ; copy the summaries you want judged into profiles/.
set -u
WL=${1:-rmat10m_biased}
TAG=${2:-r02}
OUT=gpurun_out/prof_${TAG}_${WL}
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 bench.py --workload $WL --steps 3 --warmup 2 --no-cpu-baseline --steady"   # (--steady: every launch of the walk kernel is a steady-state one)
echo "== stats pass" | tee "$OUT/log.txt"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- $BENCH > "$OUT/bench_stats.json" 2>> "$OUT/log.txt" || echo "stats pass rc=$?" >> "$OUT/log.txt"
for PASS in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "GRBM_GUI_ACTIVE GRBM_COUNT" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"; do
  NAME=$(echo $PASS | tr ' ' '_' | cut -c1-40)
  echo "== pmc pass: $PASS" >> "$OUT/log.txt"
  timeout -k 10 400 rocprofv3 --pmc $PASS --output-format csv -d "$OUT/pmc_$NAME" -o pmc -- $BENCH > /dev/null 2>> "$OUT/log.txt" || echo "pmc pass $NAME rc=$?" >> "$OUT/log.txt"
done
find "$OUT" -name '*.csv' | head -50 >> "$OUT/log.txt"
python3 scripts/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
