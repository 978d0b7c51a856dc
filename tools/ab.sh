#!/bin/bash
export GRANDPLUS_SYNTH_CACHE=${GRANDPLUS_SYNTH_CACHE:-/dev/shm/gp_synth}
# A/B builds of the HIP library on the SAME GPU box (boxes differ by a few % in clocks):
#   tools/ab.sh "liba.so libb.so ..." [workload ...]      (ROWS=16384 STEPS=3 by default)
# Build the variants with tools/build_variant.sh <name> from the source state under test.
LIBS=$1; shift
W=${@:-mag reddit pubmed cora amazon2m}
ROWS=${ROWS:-16384}; STEPS=${STEPS:-3}
for w in $W; do
  r=$ROWS; [ $w = amazon2m ] && r=4096
  for rep in 1 2; do for lib in $LIBS; do
    GRANDPLUS_LIB=$lib python bench.py --workload $w --seeds-per-gpu $r --steps $STEPS --warmup 2 --no-cpu-baseline $BENCH_EXTRA 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$w $lib', round(d['value']), d['roofline']['kernel_ms_avg'])"
  done; done
done
