#!/bin/bash
# Runs ON THE GPU BOX (round 3, first call): probes + the full GPU tier + a bench line.
set -o pipefail
O=gpurun_out/r03c1; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -o /tmp/alu_probe2 scripts/alu_probe2.hip && timeout -k 5 300 /tmp/alu_probe2 > $O/alu_probe2.txt 2>&1
echo "alu_probe2 rc=$?"
hipcc --offload-arch=gfx950 -O3 -o /tmp/machine_probe scripts/machine_probe.hip && timeout -k 5 300 /tmp/machine_probe 4096 > $O/machine_probe.txt 2>&1
echo "machine_probe rc=$?"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --workload rmat1m_biased --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_gloo2.json 2> $O/bench_gloo2.err; echo "bench2 rc=$?"
cat $O/bench_gloo2.json | head -c 3000
