#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 stats + PMC passes for every bench workload, condensed into
# gpurun_out/collected/ (the raw traces are too big to travel back), then the bench lines themselves with CPU
# baselines.  Afterwards, here: cp gpurun_out/collected/* profiles/
# usage: scripts/regen_all.sh [tag] ["workloads to profile" (may be "")] [profiles-only]   (a gpurun call lasts 20 minutes
# at most: profile the workloads in two or three calls with `profiles-only`, copying pmc_traffic.json into profiles/
# between them, then run the bench lines and probes with an empty workload list)
TAG=${1:-r04}
WLS=${2-"rmat10m_biased rmat10m_uniform er100k_uniform kg5m_triples elist_rmat1m_uniform elist_rmat1m_biased windows_1m walk_windows_1m"}
mkdir -p gpurun_out/collected
for w in $WLS; do
  bash scripts/profile_gpu.sh $w $TAG > gpurun_out/collected/prof_log_$w.txt 2>&1
  python3 scripts/collect_profiles.py $TAG $w --dst gpurun_out/collected > /dev/null
  rm -rf gpurun_out/prof_${TAG}_$w
  echo "profiled $w"
done
cp gpurun_out/collected/pmc_traffic.json profiles/pmc_traffic.json   # bench.py reads traffic / misses from here
if [ "${3:-}" = "profiles-only" ]; then du -sh gpurun_out; exit 0; fi
for w in ${BENCH_WLS:-default rmat10m_uniform er100k_uniform papers100m_uniform papers100m_biased kg5m_triples elist_rmat1m_uniform elist_rmat1m_biased windows_1m walk_windows_1m}; do
  if [ $w = default ]; then ARG=""; else ARG="--workload $w"; fi
  timeout -k 10 300 python3 bench.py $ARG > gpurun_out/collected/${TAG}_bench_$w.json 2> gpurun_out/collected/bench_$w.err || echo "bench $w failed"
  echo "bench $w done"
done
rm -f gpurun_out/collected/prof_log_*.txt
if [ "${3:-}" = "bench-only" ]; then du -sh gpurun_out; exit 0; fi
# the machine's yardsticks and the side measurements DESIGN.md quotes
hipcc --offload-arch=gfx950 -O3 -o /tmp/machine_probe scripts/machine_probe.hip && timeout -k 5 300 /tmp/machine_probe 4096 > gpurun_out/collected/${TAG}_machine_probe.txt 2>&1
hipcc --offload-arch=gfx950 -O3 -o /tmp/alu_probe2 scripts/alu_probe2.hip && timeout -k 5 300 /tmp/alu_probe2 > gpurun_out/collected/${TAG}_alu_probe2.txt 2>&1
timeout -k 10 300 python3 scripts/time_small_batch.py 2>&1 | grep -v amdgpu.ids > gpurun_out/collected/${TAG}_small_batches.txt
timeout -k 10 300 python3 scripts/time_windows.py 2>&1 | grep -v amdgpu.ids > gpurun_out/collected/${TAG}_time_windows.txt
timeout -k 10 300 python3 scripts/tune_biased.py --what 14 --runs 5 --count 2>&1 | grep -v amdgpu.ids > gpurun_out/collected/${TAG}_pool_phases.txt
# the edge-list pool: phases of the bench batch (one walker per node), and of a batch four times as long (the drain of the
# pools at the end of a batch is a quarter as heavy there)
{ timeout -k 10 300 python3 scripts/tune_elist.py --runs 5 --count --tag "1 walker per node" 2>&1; timeout -k 10 300 python3 scripts/tune_elist.py --runs 4 --count --walkers-mult 4 --tag "4 walkers per node" 2>&1; } | grep -v amdgpu.ids > gpurun_out/collected/${TAG}_elist_pool_phases.txt
du -sh gpurun_out
