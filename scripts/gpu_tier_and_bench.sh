#!/bin/bash
# Runs ON THE GPU BOX: the whole GPU tier, then bench lines given as arguments (workload names).
set -o pipefail
O=gpurun_out/tier; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
for WL in "$@"; do
  timeout -k 10 300 python bench.py --workload $WL > $O/bench_$WL.json 2> $O/bench_$WL.err; echo "bench $WL rc=$?"
  python - $O/bench_$WL.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d["roofline"]
    print("  step %.3f ms kernel %.3f finish %.3f [%s] value %.3e frac %.3f cpu %s" % (d["ms_per_step"], r["kernel_ms"], r["finish_ms"], r["kernel"], d["value"], r["frac"], d["cpu_baseline"] and "%.3e" % d["cpu_baseline"]["value"]))
except Exception as e:
    print("  no line:", e)
PY
done
