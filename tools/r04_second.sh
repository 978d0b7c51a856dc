#!/bin/bash
export GRANDPLUS_SYNTH_CACHE=/dev/shm/gp_synth
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_sketch.py -x -q 2>&1 | tail -40 > gpurun_out/sk_tests.txt
cat gpurun_out/sk_tests.txt
timeout 600 python tools/sk_quick.py mag 65536 > gpurun_out/sk_quick_mag.txt 2>&1
cat gpurun_out/sk_quick_mag.txt | tail -5
