#!/bin/bash
export GRANDPLUS_SYNTH_CACHE=/dev/shm/gp_synth
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > gpurun_out/gpu_tests.txt
tail -6 gpurun_out/gpu_tests.txt
timeout 600 python tools/sk_quick.py mag 65536 > gpurun_out/sk_quick_mag.txt 2>&1
tail -4 gpurun_out/sk_quick_mag.txt
