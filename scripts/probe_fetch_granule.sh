#!/bin/bash
# GPU box: how many HBM bytes does ONE L2 miss of a random gather move?  Runs scripts/machine_probe.hip's single
# launch (a known number of dependent random gathers over 4 GB) under rocprofv3 PMC passes and divides.
# Output: gpurun_out/fetch_granule.txt (copy to profiles/).
set -u
export TMPDIR=/tmp
OUT=gpurun_out/fetch_granule
mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $OUT/machine_probe scripts/machine_probe.hip || exit 1
{
for BYTES in 4 16 32; do
  for CNT in FETCH_SIZE TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum; do
    rm -rf $OUT/p
    timeout -k 10 200 rocprofv3 --pmc $CNT --output-format csv -d $OUT/p -o pmc -- $OUT/machine_probe one 4096 $BYTES > $OUT/run.txt 2> $OUT/err.txt
    G=$(grep '^one:' $OUT/run.txt | sed 's/.*timed launch = \([0-9]*\) gathers.*/\1/')
    python3 - "$OUT/p" "$CNT" "$BYTES" "$G" <<'P'
import csv, glob, sys
d, cnt, b, g = sys.argv[1], sys.argv[2], int(sys.argv[3]), float(sys.argv[4])
best = 0.0
for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        if "gather_kernel" in row["Kernel_Name"]:
            best = max(best, float(row["Counter_Value"]))   # the timed launch is the largest dispatch
unit = 1024.0 if cnt == "FETCH_SIZE" else 1.0
print("%2d-byte gathers  %-22s = %.4g  -> %.2f %s per gather (%.4g gathers)" % (
    b, cnt, best, best * unit / g, "bytes" if cnt == "FETCH_SIZE" else "events", g))
P
  done
done
} | tee gpurun_out/fetch_granule.txt
