#!/bin/bash
# Instruction counts per phase, by difference (diagnostic build): run 0 = everything, 1 = TOP-K skipped,
# 2 = EXPAND executed twice (second pass adds zeros).  SCAN + row/level overhead = run 1 - (run 2 - run 0).
#   tools/instr_split.sh [workload]
W=${1:-mag}; OUT=gpurun_out/instr_split
export TMPDIR=/tmp GRANDPLUS_DIAG=1
mkdir -p $OUT
for f in ${FLAGS:-0 1 2}; do
  timeout -k 5 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/f$f -- python3 bench.py --workload $W --seeds-per-gpu 16384 --steps 3 --warmup 1 --no-cpu-baseline --no-host-api --no-next-rows --diag-flags $f > $OUT/f$f.log 2>&1
  echo "diag_flags=$f: $(python tools/pmc_summary.py $OUT/f$f $OUT/f$f.json --rows 16384 --warmup 1 | grep per_row | tr -d '\n')"
done
