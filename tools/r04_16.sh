#!/bin/bash
export GRANDPLUS_SYNTH_CACHE=/dev/shm/gp_synth
mkdir -p gpurun_out
( timeout 600 python tools/sk_quick.py amazon2m 12350; timeout 300 python tools/sk_quick.py pubmed 65536; timeout 300 python tools/sk_quick.py reddit 65536 ) 2>&1 | grep -v amdgpu.ids > gpurun_out/sk_other.txt
cat gpurun_out/sk_other.txt
