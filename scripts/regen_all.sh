#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 stats + PMC passes for every bench workload, condensed into
# gpurun_out/collected/ (the raw traces are too big to travel back), then the bench lines themselves with CPU
# baselines.  Afterwards, here: cp gpurun_out/collected/* profiles/
TAG=${1:-r02}
WLS=${2:-"rmat10m_biased rmat10m_uniform er100k_uniform kg5m_triples elist_rmat1m_uniform elist_rmat1m_biased windows_1m"}
mkdir -p gpurun_out/collected
for w in $WLS; do
  bash scripts/profile_gpu.sh $w $TAG > gpurun_out/collected/prof_log_$w.txt 2>&1
  python3 scripts/collect_profiles.py $TAG $w --dst gpurun_out/collected > /dev/null
  rm -rf gpurun_out/prof_${TAG}_$w
  echo "profiled $w"
done
cp gpurun_out/collected/pmc_traffic.json profiles/pmc_traffic.json   # bench.py reads traffic / misses from here
for w in default rmat10m_uniform er100k_uniform papers100m_uniform papers100m_biased kg5m_triples elist_rmat1m_uniform elist_rmat1m_biased windows_1m; do
  if [ $w = default ]; then ARG=""; else ARG="--workload $w"; fi
  timeout -k 10 300 python3 bench.py $ARG > gpurun_out/collected/${TAG}_bench_$w.json 2> gpurun_out/collected/bench_$w.err || echo "bench $w failed"
  echo "bench $w done"
done
rm -f gpurun_out/collected/prof_log_*.txt
du -sh gpurun_out
