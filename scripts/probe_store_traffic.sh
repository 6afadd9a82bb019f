#!/bin/bash
# GPU box: HBM bytes fetched / written per byte of walk output, for each way of storing the columns
# (scripts/store_probe.hip).  Output: gpurun_out/store_traffic.txt (copy to profiles/).
set -u
export TMPDIR=/tmp
OUT=gpurun_out/store_traffic
mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $OUT/store_probe scripts/store_probe.hip || exit 1
WALKERS=${1:-4000000}
{
for MODE in 0 1 2 3 4; do
  for CNT in FETCH_SIZE WRITE_SIZE; do
    rm -rf $OUT/p
    timeout -k 10 200 rocprofv3 --pmc $CNT --output-format csv -d $OUT/p -o pmc -- $OUT/store_probe $MODE $WALKERS 81 2 > $OUT/run.txt 2> $OUT/err.txt
    python3 - "$OUT/p" "$CNT" "$MODE" "$WALKERS" "$(grep '^store:' $OUT/run.txt)" <<'P'
import csv, glob, sys
d, cnt, mode, walkers, line = sys.argv[1], sys.argv[2], int(sys.argv[3]), float(sys.argv[4]), sys.argv[5]
vals = []
for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        if "store_kernel" in row["Kernel_Name"]:
            vals.append(float(row["Counter_Value"]))
v = sum(vals) / max(1, len(vals)) * 1024.0
out_bytes = walkers * 81 * 8
gather_bytes = walkers * 81 * 2 * 64
print("mode %d %-10s = %.1f GB per launch; minus the gathers' %.1f GB: %.1f GB = %.2f x the %.2f GB of output   | %s" % (
    mode, cnt, v / 1e9, gather_bytes / 1e9 if cnt == "FETCH_SIZE" else 0.0,
    (v - (gather_bytes if cnt == "FETCH_SIZE" else 0)) / 1e9, (v - (gather_bytes if cnt == "FETCH_SIZE" else 0)) / out_bytes,
    out_bytes / 1e9, line[:60] + line[-40:]))
P
  done
done
} | tee gpurun_out/store_traffic.txt
