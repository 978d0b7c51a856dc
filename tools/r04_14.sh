#!/bin/bash
export GRANDPLUS_SYNTH_CACHE=/dev/shm/gp_synth
timeout 600 python -m pytest tests/test_gpu_sketch.py -q -x -k flat 2>&1 | grep -E "AssertionError|assert|passed|failed" | head
