#!/bin/bash
# Runs ON THE GPU BOX: bench the default workload once per prebuilt libtrw_hip.so variant in variants/.
# usage: scripts/try_variants.sh [workload] ; prints one line per variant
WL=${1:-rmat10m_biased}
cp torch_random_walk_amd/libtrw_hip.so /tmp/libtrw_orig.so
for f in variants/*.so; do
  cp "$f" torch_random_walk_amd/libtrw_hip.so
  timeout -k 10 200 python bench.py --workload $WL --no-cpu-baseline > /tmp/v.json 2> /tmp/v.err || { echo "$f FAILED"; tail -3 /tmp/v.err; continue; }
  python - "$f" <<'PY'
import json, sys
d = json.load(open("/tmp/v.json")); r = d["roofline"]
print("VAR %-28s step %.2f ms  kernel %.2f  prepare %.2f  finish %.2f" % (sys.argv[1], d["ms_per_step"], r["kernel_ms"], r["prepare_ms"], r["finish_ms"]))
PY
done
cp /tmp/libtrw_orig.so torch_random_walk_amd/libtrw_hip.so
