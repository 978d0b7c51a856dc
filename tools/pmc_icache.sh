#!/bin/bash
# Instruction-cache behaviour of the GFPush kernel (is the 130 KB kernel fetch-bound?).
W=${1:-mag}; OUT=${2:-gpurun_out/pmci}; shift 2
export TMPDIR=/tmp
mkdir -p $OUT
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQC_ICACHE_BUSY_CYCLES SQC_TC_INST_REQ SQC_TC_STALL SQC_DCACHE_MISSES"; do
  i=$((i+1))
  timeout -k 5 100 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python3 bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-host-api --no-next-rows "$@" > $OUT/g$i.log 2>&1
  echo "group $i ($grp): rc=$?"
done
python3 - <<PY
import csv, glob, os
from collections import defaultdict
per = {}
for f in sorted(glob.glob(os.path.join("$OUT", "**", "*counter_collection.csv"), recursive=True)):
    by = defaultdict(lambda: defaultdict(float))
    for r in csv.DictReader(open(f)):
        if "gfpush_kernel<" not in r["Kernel_Name"]: continue
        by[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    for c, d in by.items():
        vals = [d[k] for k in sorted(d)][1:]
        if vals: per[c] = sum(vals) / len(vals)
for k, v in per.items(): print(f"{k:28s} {v:.4g}")
PY
