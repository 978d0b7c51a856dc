#!/bin/bash
export GRANDPLUS_SYNTH_CACHE=/dev/shm/gp_synth
export TMPDIR=/tmp
mkdir -p gpurun_out/r04a
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04a/trace -- python3 bench.py --prewarm 0 --steps 5 --warmup 2 --no-cpu-baseline --no-host-api --no-next-rows > gpurun_out/r04a/trace.log 2>&1
cp $(ls -t gpurun_out/r04a/trace/*/*kernel_stats.csv | head -1) gpurun_out/r04a/kernel_stats.csv
cat gpurun_out/r04a/kernel_stats.csv | head -8
tail -2 gpurun_out/r04a/trace.log
rm -rf gpurun_out/r04a/trace
