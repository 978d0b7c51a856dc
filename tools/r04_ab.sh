#!/bin/bash
# A/B of library builds on ONE box (boxes differ by +-6 %): tools/r04_ab.sh "<lib> <lib> ..." [rounds] [sk_quick options]
# Each round runs every library once (tools/sk_quick.py mag 65536, the sketch-kernel line); libraries live in grand_plus_amd/.
LIBS=${1:-"libgrandplus.so libgrandplus.so.new"}; ROUNDS=${2:-3}; shift 2
export GRANDPLUS_SYNTH_CACHE=/dev/shm/gp_synth
export SKQ_ONLY="sketch 768"
mkdir -p gpurun_out
: > gpurun_out/sk_ab.txt
for r in $(seq 1 $ROUNDS); do
  for l in $LIBS; do
    echo -n "$l: " >> gpurun_out/sk_ab.txt
    GRANDPLUS_LIB=$l timeout 300 python tools/sk_quick.py mag 65536 "$@" 2>&1 | grep "sketch 768 " | cut -c1-150 >> gpurun_out/sk_ab.txt
  done
done
cat gpurun_out/sk_ab.txt
