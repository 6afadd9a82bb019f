#!/bin/bash
# Runs ON THE GPU BOX: the prepared-graph suites, then prepare / walk times at the three marking levels.
set -o pipefail
O=gpurun_out/r03prep; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_prepared.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for w in 14 6 2; do
  timeout -k 10 300 python scripts/tune_biased.py --what $w --runs 3 2>&1 | grep -v amdgpu | cut -c1-100
done
