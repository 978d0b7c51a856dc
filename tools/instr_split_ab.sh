#!/bin/bash
# Per-phase instruction counts (by difference, see instr_split.sh) of several -DGP_DIAG builds: tools/instr_split_ab.sh workload lib...
W=$1; shift
export TMPDIR=/tmp
for lib in "$@"; do for f in 0 1 2; do
  OUT=gpurun_out/isab/$lib.$f; rm -rf $OUT; mkdir -p $OUT
  GRANDPLUS_LIB=$lib timeout -k 5 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT -- python3 bench.py --workload $W --seeds-per-gpu 16384 --steps 3 --warmup 1 --no-cpu-baseline --no-host-api --no-next-rows --diag-flags $f > $OUT.log 2>&1
  echo "$lib diag_flags=$f: $(python tools/pmc_summary.py $OUT $OUT.json --rows 16384 --warmup 1 | grep per_row | tr -d '\n')"
done; done
