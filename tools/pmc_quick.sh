#!/bin/bash
# Quick utilisation counters of the GFPush kernel (4 rocprofv3 --pmc passes, one counter group each).
#   tools/pmc_quick.sh [workload] [out_dir] [extra bench args...]
W=${1:-mag}; OUT=${2:-gpurun_out/pmcq}; shift 2
export TMPDIR=/tmp
mkdir -p $OUT
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT"; do
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
        if "gfpush_kernel" not in r["Kernel_Name"]: continue
        by[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    for c, d in by.items():
        vals = [d[k] for k in sorted(d)]
        # two launches per call (main + retry): keep the big ones
        big = [v for v in vals if v > 0.05 * max(vals)][1:]
        if big: per[c] = sum(big) / len(big)
for k, v in per.items(): print(f"{k:28s} {v:.4g}")
PY
