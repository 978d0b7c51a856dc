#!/bin/bash
# A/B of library builds for the GENERAL kernel's lines on one box: tools/r04_general_ab.sh "<lib> ..." "<workload:rows> ..."
LIBS=$1; WL=${2:-"pubmed:65536 cora:65536 amazon2m:12350"}
export GRANDPLUS_SYNTH_CACHE=/dev/shm/gp_synth SKQ_ONLY="general"
mkdir -p gpurun_out; : > gpurun_out/gen_ab.txt
for r in 1 2; do for w in $WL; do for l in $LIBS; do
  echo -n "$l: " >> gpurun_out/gen_ab.txt
  GRANDPLUS_LIB=$l timeout 600 python tools/sk_quick.py ${w%%:*} ${w##*:} 2>&1 | grep " general " | sed 's/; retried.*//' | cut -c1-130 >> gpurun_out/gen_ab.txt
done; done; done; cat gpurun_out/gen_ab.txt
