#!/bin/bash
export GRANDPLUS_SYNTH_CACHE=/dev/shm/gp_synth
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > gpurun_out/gpu_tests.txt
tail -15 gpurun_out/gpu_tests.txt
