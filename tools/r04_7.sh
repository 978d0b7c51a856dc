#!/bin/bash
export GRANDPLUS_SYNTH_CACHE=/dev/shm/gp_synth
mkdir -p gpurun_out
GRANDPLUS_LIB=libgrandplus_skt.so timeout 600 python tools/sk_phases.py mag 65536 sk_block_threads=768 sk_r_atomic=0 > gpurun_out/sk_phases_mag.txt 2>&1
GRANDPLUS_LIB=libgrandplus_skt.so timeout 600 python tools/sk_phases.py mag 4096 sk_block_threads=768 sk_r_atomic=0 max_workgroups=8 >> gpurun_out/sk_phases_mag.txt 2>&1
cat gpurun_out/sk_phases_mag.txt
