#!/bin/bash
# A/B of library builds / options on one box: tools/r04_ab.sh  (edit the list below)
export GRANDPLUS_SYNTH_CACHE=/dev/shm/gp_synth
mkdir -p gpurun_out
export SKQ_ONLY="sketch 768"
( timeout 300 python tools/sk_quick.py mag 65536; GRANDPLUS_LIB=libgrandplus.so.new timeout 300 python tools/sk_quick.py mag 65536 sk_block_threads=1024; timeout 300 python tools/sk_quick.py mag 65536; GRANDPLUS_LIB=libgrandplus.so.new timeout 300 python tools/sk_quick.py mag 65536 sk_block_threads=1024 sk_lg_mu=13 ) 2>&1 | grep "sketch 768 " | cut -c1-170 > gpurun_out/sk_ab.txt
cat gpurun_out/sk_ab.txt
