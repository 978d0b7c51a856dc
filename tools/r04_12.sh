#!/bin/bash
export GRANDPLUS_SYNTH_CACHE=/dev/shm/gp_synth
export GRANDPLUS_DEBUG_MERGE=1
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -s -k "random_digraphs" 2>&1 | grep -i "merge\|passed\|failed" | sort | uniq -c | sort -rn | head -30
