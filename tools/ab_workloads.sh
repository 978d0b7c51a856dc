#!/bin/bash
# A/B of library builds over several workloads on ONE box: tools/ab_workloads.sh "<lib> <lib> ..." "<workload>:<variant>:<rows> ..." [rounds]
# variant = general | "sketch 768" (spaces as _), e.g. "mag:sketch_768:65536 pubmed:general:65536 amazon2m:general:12350"
LIBS=${1:-"libgrandplus.so"}; CASES=${2:-"mag:sketch_768:65536"}; ROUNDS=${3:-2}
export GRANDPLUS_SYNTH_CACHE=${GRANDPLUS_SYNTH_CACHE:-/dev/shm/gp_synth}
mkdir -p gpurun_out $GRANDPLUS_SYNTH_CACHE
: > gpurun_out/ab_workloads.txt
for c in $CASES; do
  IFS=: read -r W V S <<< "$c"
  for r in $(seq 1 $ROUNDS); do
    for l in $LIBS; do
      echo -n "$l: " >> gpurun_out/ab_workloads.txt
      SKQ_ONLY="${V//_/ }" GRANDPLUS_LIB=$l timeout 400 python tools/sk_quick.py $W $S 2>&1 | grep " best " | cut -c1-150 >> gpurun_out/ab_workloads.txt
    done
  done
done
cat gpurun_out/ab_workloads.txt
