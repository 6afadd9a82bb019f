#!/bin/bash
# Runs ON THE GPU BOX: like try_variants.sh, but every variant first has to pass the biased parity tests
# (a short timeout guards against a variant that hangs).
WL=${1:-rmat10m_biased}
cp torch_random_walk_amd/libtrw_hip.so /tmp/libtrw_orig.so
for f in variants/*.so; do
  cp "$f" torch_random_walk_amd/libtrw_hip.so
  if ! timeout -k 5 120 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "variants or biased or fuzz or hub" > /tmp/vt.log 2>&1; then echo "$f PARITY FAILED"; tail -3 /tmp/vt.log; continue; fi
  timeout -k 10 120 python bench.py --workload $WL --no-cpu-baseline > /tmp/v.json 2> /tmp/v.err || { echo "$f BENCH FAILED"; tail -3 /tmp/v.err; continue; }
  python - "$f" <<'PY'
import json, sys
d = json.load(open("/tmp/v.json")); r = d["roofline"]
print("VAR %-28s step %.2f ms  kernel %.2f  prepare %.2f" % (sys.argv[1], d["ms_per_step"], r["kernel_ms"], r["prepare_ms"]))
PY
done
cp /tmp/libtrw_orig.so torch_random_walk_amd/libtrw_hip.so
