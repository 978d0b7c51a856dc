#!/bin/bash
# HBM-side traffic of the GFPush kernel per launch (separate FETCH_SIZE / WRITE_SIZE passes, as the guide prescribes).
#   GRANDPLUS_LIB=... tools/traffic.sh [workload] [out_dir]
W=${1:-mag}; OUT=${2:-gpurun_out/traffic}; shift 2
export TMPDIR=/tmp
mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 100 rocprofv3 --pmc $c --output-format csv -d $OUT/$c -- python3 bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-host-api --no-next-rows "$@" > $OUT/$c.log 2>&1
done
python3 - <<PY
import csv, glob, os
from collections import defaultdict
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    by = defaultdict(float)
    for f in glob.glob(os.path.join("$OUT", c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "gfpush_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c: by[int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    vals = [by[k] for k in sorted(by)]
    big = [v for v in vals if v > 0.05 * max(vals)][1:]
    print(c, "GB per launch (raw, KB counter x 1024):", round(sum(big) / len(big) * 1024 / 1e9, 2))
PY
