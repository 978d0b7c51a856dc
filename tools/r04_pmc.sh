#!/bin/bash
export GRANDPLUS_SYNTH_CACHE=/dev/shm/gp_synth
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
mkdir -p gpurun_out/r04a
tools/collect_pmc.sh mag gpurun_out/r04a/pmc > gpurun_out/r04a/pmc.log 2>&1
python tools/pmc_summary.py gpurun_out/r04a/pmc gpurun_out/r04a/pmc_summary.json
rm -rf gpurun_out/r04a/pmc/*/*/*.db 2>/dev/null
du -sh gpurun_out/r04a
