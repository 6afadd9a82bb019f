#!/bin/bash
# Runs ON THE GPU BOX: bench.py's N > 1 control flow with 2..4 gloo ranks on the one card (the numbers mean nothing; what
# is checked is that every configuration prints its line and passes bench.py's own checks of the gathered blocks).
# `--backend nccl` (on a node with as many GPUs as the largest --gpus below): the same matrix over RCCL, one rank per GPU.
BACKEND=gloo
if [ "$1" = "--backend" ]; then BACKEND=$2; fi
run() {
  timeout 300 python bench.py "$@" --backend $BACKEND --steps 2 --warmup 1 --no-cpu-baseline > /tmp/o.json 2>/tmp/o.err; rc=$?
  python - "$*" $rc <<'PY'
import json, sys
lines = [ln for ln in open("/tmp/o.json") if ln.startswith("{")]
if sys.argv[2] == "0" and len(lines) == 1:
    d = json.loads(lines[0]); m = d.get("multi_gpu") or {}
    print("OK   %-62s n_gpus %d value %.2e wire %s chunks %s gather_ms %s" % (sys.argv[1], d["n_gpus"], d["value"], m.get("wire_dtype"), m.get("gather_chunks"), m.get("gather_ms") and round(m["gather_ms"], 1)))
else:
    print("FAIL %-62s rc=%s" % (sys.argv[1], sys.argv[2])); print(open("/tmp/o.err").read()[-1500:])
PY
}
run --gpus 3 --workload rmat1m_biased
run --gpus 4 --workload rmat1m_biased --gather-chunks 7
run --gpus 2 --workload rmat1m_biased --wire int64 --gather-chunks 1
run --gpus 3 --workload rmat1m_uniform
run --gpus 2 --workload kg5m_triples
run --gpus 2 --workload elist_rmat1m_biased
run --gpus 2 --workload walk_windows_1m
run --gpus 2 --workload windows_1m
run --gpus 2 --workload rmat1m_biased --no-gather
run --gpus 2 --workload rmat1m_biased --result int32
run --gpus 3 --workload rmat1m_biased --result int32 --gather-chunks 3
run --gpus 2 --workload rmat1m_uniform --result int32
