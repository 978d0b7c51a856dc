#!/bin/bash
# A/B of sketch-kernel OPTIONS with one library on one box: tools/r04_opts_ab.sh <lib> "<options>" "<options>" ...  (three rounds each)
LIB=$1; shift
export GRANDPLUS_SYNTH_CACHE=/dev/shm/gp_synth SKQ_ONLY="sketch 768" GRANDPLUS_LIB=$LIB
W=${AB_WORKLOAD:-mag}
mkdir -p gpurun_out; : > gpurun_out/sk_ab.txt
for r in 1 2 3; do for o in "$@"; do
  echo -n "[$o] " >> gpurun_out/sk_ab.txt
  timeout 300 python tools/sk_quick.py $W 65536 $o 2>&1 | grep "sketch 768 " | sed 's/.*wgs 512//' | cut -c1-110 >> gpurun_out/sk_ab.txt
done; done; cat gpurun_out/sk_ab.txt
