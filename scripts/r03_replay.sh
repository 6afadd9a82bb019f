#!/bin/bash
set -o pipefail
O=gpurun_out/r03rp; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_prepared.py tests/test_edge_list.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout -k 10 300 python scripts/time_small_batch.py 2>&1 | grep -v amdgpu.ids | head -13 | tee $O/small_batches.txt
