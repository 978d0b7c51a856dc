#!/bin/bash
export GRANDPLUS_SYNTH_CACHE=/dev/shm/gp_synth
mkdir -p gpurun_out
export SKQ_ONLY="sketch 768"
( timeout 300 python tools/sk_quick.py mag 65536; timeout 300 python tools/sk_quick.py mag 65536 sk_direct_max=300; timeout 300 python tools/sk_quick.py mag 65536 sk_direct_max=700; timeout 300 python tools/sk_quick.py mag 65536 sk_direct_max=1100;  timeout 300 python tools/sk_quick.py mag 65536 solo_levels=0; timeout 300 python tools/sk_quick.py mag 65536 ) 2>&1 | grep "sketch 768 " > gpurun_out/sk_ab.txt
cat gpurun_out/sk_ab.txt
