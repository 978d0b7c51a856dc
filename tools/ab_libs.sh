#!/bin/bash
# A/B of library builds on ONE box (boxes differ by +-6 %): tools/ab_libs.sh "<lib> <lib> ..." [rounds] [workload] [sk_quick options]
# Each round runs every library once (tools/sk_quick.py <workload> 65536); libraries live in grand_plus_amd/ (tools/build_variant.sh).
LIBS=${1:-"libgrandplus.so"}; ROUNDS=${2:-3}; W=${3:-mag}; shift; shift; shift
export GRANDPLUS_SYNTH_CACHE=${GRANDPLUS_SYNTH_CACHE:-/dev/shm/gp_synth}
export SKQ_ONLY=${SKQ_ONLY:-"sketch 768"}
mkdir -p gpurun_out $GRANDPLUS_SYNTH_CACHE
: > gpurun_out/ab_libs.txt
for r in $(seq 1 $ROUNDS); do
  for l in $LIBS; do
    echo -n "$l: " >> gpurun_out/ab_libs.txt
    GRANDPLUS_LIB=$l timeout 300 python tools/sk_quick.py $W 65536 "$@" 2>&1 | grep -v "host API" | grep " best " | cut -c1-160 >> gpurun_out/ab_libs.txt
  done
done
cat gpurun_out/ab_libs.txt
