#!/bin/bash
export GRANDPLUS_SYNTH_CACHE=/dev/shm/gp_synth
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_sketch.py -q -x 2>&1 | tail -3
export SKQ_ONLY="sketch 768"
( timeout 300 python tools/sk_quick.py mag 65536; timeout 300 python tools/sk_quick.py mag 65536 sk_lg_mr=12; timeout 300 python tools/sk_quick.py mag 65536 sk_target=256; timeout 300 python tools/sk_quick.py mag 65536 sk_target=64; timeout 300 python tools/sk_quick.py mag 65536 sk_lg_mu=12 ) 2>&1 | grep "sketch 768 " > gpurun_out/sk_ab.txt
cat gpurun_out/sk_ab.txt
GRANDPLUS_LIB=libgrandplus_skt.so timeout 600 python tools/sk_phases.py mag 65536 sk_block_threads=768 > gpurun_out/sk_phases_mag.txt 2>&1
head -16 gpurun_out/sk_phases_mag.txt
