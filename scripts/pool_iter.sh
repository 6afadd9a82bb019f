#!/bin/bash
# Runs ON THE GPU BOX: one iteration of the walker-pool tuning loop -- parity suites, the phase counters, the bench line.
set -o pipefail
O=gpurun_out/pool_iter; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_prepared.py tests/test_gpu_fuzz.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
timeout -k 10 300 python scripts/tune_biased.py --what 14 --runs 4 --count 2>&1 | tail -6
timeout -k 10 300 python bench.py --no-cpu-baseline --steady --steps 8 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - $O/bench.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print("step %.2f ms  kernel %.2f  [%s]" % (d["ms_per_step"], r["kernel_ms"], r["kernel"]))
PY
