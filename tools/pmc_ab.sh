#!/bin/bash
# usage: pmc_ab.sh workload lib...
W=$1; shift
export TMPDIR=/tmp
for lib in "$@"; do
  OUT=gpurun_out/pmcab/$lib; mkdir -p $OUT
  GRANDPLUS_LIB=$lib timeout -k 5 100 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT -- python3 bench.py --workload $W --seeds-per-gpu 16384 --steps 3 --warmup 1 --no-cpu-baseline --no-host-api --no-next-rows > $OUT.log 2>&1
  python3 - <<PY
import csv, glob, os
from collections import defaultdict
per = {}
for f in sorted(glob.glob(os.path.join("$OUT", "**", "*counter_collection.csv"), recursive=True)):
    by = defaultdict(lambda: defaultdict(float))
    for r in csv.DictReader(open(f)):
        if "gfpush_kernel" not in r["Kernel_Name"]: continue
        by[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    for c, d in by.items():
        vals = [d[k] for k in sorted(d)]
        big = [v for v in vals if v > 0.05 * max(vals)][1:]
        if big: per[c] = sum(big) / len(big) / 16384
print("$W $lib", {k: round(v) for k, v in per.items()})
PY
done
