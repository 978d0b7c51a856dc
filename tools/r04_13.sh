#!/bin/bash
export GRANDPLUS_SYNTH_CACHE=/dev/shm/gp_synth
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_sketch.py -q -x 2>&1 | tail -12
timeout 600 python tools/sk_quick.py mag 65536 > gpurun_out/sk_quick_mag.txt 2>&1
tail -4 gpurun_out/sk_quick_mag.txt
