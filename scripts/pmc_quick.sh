#!/bin/bash
# quick SQ/TCC counter passes for one workload: scripts/pmc_quick.sh <workload> <tag>
WL=${1:-rmat10m_biased}; TAG=${2:-q}
OUT=gpurun_out/prof_${TAG}_${WL}; mkdir -p "$OUT"; export TMPDIR=/tmp
BENCH="python3 bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline"
for PASS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  NAME=$(echo $PASS | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --pmc $PASS --output-format csv -d "$OUT/pmc_$NAME" -o pmc -- $BENCH > /dev/null 2>> "$OUT/log.txt" || echo "pass $NAME rc=$?" >> "$OUT/log.txt"
done
python3 scripts/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
