#!/bin/bash
export GRANDPLUS_SYNTH_CACHE=/dev/shm/gp_synth
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_sketch.py -q -x 2>&1 | tail -30 > gpurun_out/sk_tests.txt
tail -8 gpurun_out/sk_tests.txt
timeout 600 python tools/sk_quick.py mag 65536 > gpurun_out/sk_quick_mag.txt 2>&1
tail -3 gpurun_out/sk_quick_mag.txt
GRANDPLUS_LIB=libgrandplus_skt.so timeout 600 python tools/sk_phases.py mag 65536 sk_block_threads=768 > gpurun_out/sk_phases_mag.txt 2>&1
GRANDPLUS_LIB=libgrandplus_skt.so timeout 600 python tools/sk_phases.py mag 65536 sk_block_threads=512 >> gpurun_out/sk_phases_mag.txt 2>&1
cat gpurun_out/sk_phases_mag.txt
