#!/bin/bash
# Runs ON THE GPU BOX: bench one workload under several settings of one env knob.
# usage: scripts/ab_env.sh <workload> <ENV_NAME> <value> [<value> ...]
WL=$1; NAME=$2; shift 2
for v in "$@"; do
  env $NAME=$v timeout -k 10 200 python bench.py --workload $WL --no-cpu-baseline > /tmp/v.json 2> /tmp/v.err || { echo "$NAME=$v FAILED"; tail -3 /tmp/v.err; continue; }
  python - "$WL $NAME=$v" <<'PY'
import json, sys
d = json.load(open("/tmp/v.json")); r = d["roofline"]
print("AB %-40s step %.3f ms  kernel %.3f  prepare %.3f  finish %.3f  [%s]" % (sys.argv[1], d["ms_per_step"], r["kernel_ms"], r["prepare_ms"], r["finish_ms"], r["kernel"]))
PY
done
