#!/bin/bash
export GRANDPLUS_SYNTH_CACHE=/dev/shm/gp_synth
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
bash tools/collect_profiles.sh r04b > gpurun_out/r04b_collect.log 2>&1
tail -8 gpurun_out/r04b_collect.log
rm -rf gpurun_out/r04b/trace gpurun_out/r04b/pmc/*/*/*.db 2>/dev/null
ls gpurun_out/r04b
