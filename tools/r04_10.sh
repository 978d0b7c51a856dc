#!/bin/bash
export GRANDPLUS_SYNTH_CACHE=/dev/shm/gp_synth
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "kat" 2>&1 | tail -40 > gpurun_out/kat.txt
cat gpurun_out/kat.txt
for i in 1 2 3; do timeout 300 python -m pytest tests/test_gpu_parity.py -q -k "kat or dtype or golden" 2>&1 | tail -3; done
