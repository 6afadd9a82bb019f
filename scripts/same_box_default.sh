#!/bin/bash
# GPU box: the default bench line and the rocprofv3 --kernel-trace --stats pass of the same command on ONE box, so that the
# line's HIP-event kernel time and the profiler's average can be compared (boxes differ by +-1 ms on this kernel).
O=$GRAFT_REPO_ROOT/gpurun_out/same_box; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
python3 bench.py --no-cpu-baseline > $O/bench_plain.json 2> $O/bench_plain.err
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o stats -- python3 bench.py --no-cpu-baseline --steady > $O/bench_under_rocprof.json 2>> $O/log.txt
python3 - <<'PY'
import csv, glob, json
o = "gpurun_out/same_box/"
a = json.loads([l for l in open(o + "bench_plain.json") if l.startswith("{")][-1])
b = json.loads([l for l in open(o + "bench_under_rocprof.json") if l.startswith("{")][-1])
print("# scripts/same_box_default.sh: python3 bench.py (default workload), then rocprofv3 --kernel-trace --stats -- python3 bench.py --steady, same box")
print("bench.py                      ms_per_step %.3f  roofline.kernel_ms (HIP events) %.3f  value %.4e  frac %.4f" % (a["ms_per_step"], a["roofline"]["kernel_ms"], a["value"], a["roofline"]["frac"]))
print("bench.py under rocprofv3      ms_per_step %.3f  roofline.kernel_ms (HIP events) %.3f" % (b["ms_per_step"], b["roofline"]["kernel_ms"]))
for f in glob.glob(o + "stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "biased_pool_kernel" in r["Name"]:
            print("rocprofv3 kernel_stats        %-48s calls %s  avg %.3f ms  min %.3f  max %.3f" % (r["Name"][:48], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6))
PY
rm -rf $O/stats/*/*trace.csv $O/stats/*trace.csv
