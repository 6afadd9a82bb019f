#!/bin/bash
# Runs ON THE GPU BOX: A/B of the walker-pool kernels (TRW_POOL_KERNEL=1 the round-2 burst, 2 the round-3 one):
# parity suites under each, then the headline workload.
set -o pipefail
O=gpurun_out/r03ab; mkdir -p $O
for v in "$@"; do
  echo "== TRW_POOL_KERNEL=$v"
  TRW_POOL_KERNEL=$v timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_prepared.py tests/test_gpu_fuzz.py -m gpu -x -q > $O/pytest_$v.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_$v.log
  TRW_POOL_KERNEL=$v timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err; echo "bench rc=$?"
  python - $O/bench_$v.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print("step %.2f ms  kernel %.2f  [%s]  cold first call %.1f" % (d["ms_per_step"], r["kernel_ms"], r["kernel"], r["cold"]["first_call_ms"]))
PY
done
