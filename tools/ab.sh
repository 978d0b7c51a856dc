#!/bin/bash
# A/B two builds of the HIP library on the SAME GPU box (boxes differ by a few % in clocks):
#   tools/ab.sh libgrandplus_a.so libgrandplus_b.so [workload ...]
# Build the variants with tools/build_variant.sh <name> from the source state under test.
A=$1; B=$2; shift 2
W=${@:-mag reddit pubmed cora amazon2m}
for w in $W; do
  extra=""; [ $w = amazon2m ] && extra="--seeds-per-gpu 4096"
  for rep in 1 2; do for lib in $A $B; do
    GRANDPLUS_LIB=$lib python bench.py --workload $w $extra --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$w $lib', round(d['value']), d['roofline']['kernel_ms_avg'])"
  done; done
done
