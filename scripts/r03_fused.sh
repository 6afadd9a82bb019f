#!/bin/bash
set -o pipefail
O=gpurun_out/r03f; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_windows.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
timeout -k 10 300 python scripts/time_windows.py 2>&1 | tail -12
