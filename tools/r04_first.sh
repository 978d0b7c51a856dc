#!/bin/bash
# Round-4 decision experiments: workgroup scaling on the round-3 shape, and how often the 4-5x slow
# phase appears in fresh MAG processes with the default workspace, a 2 GB workspace and a pre-touched workspace.
export GRANDPLUS_SYNTH_CACHE=/dev/shm/gp_synth
mkdir -p gpurun_out
python tools/exp_wg_scale.py mag > gpurun_out/r04_mag_wg_scale.txt 2>&1
tail -3 gpurun_out/r04_mag_wg_scale.txt
: > gpurun_out/r04_slow_phase.jsonl
for i in $(seq 1 ${RUNS:-12}); do
  timeout 120 python tools/slow_phase_runs.py >> gpurun_out/r04_slow_phase.jsonl 2>gpurun_out/sp_err.txt
  timeout 120 python tools/slow_phase_runs.py workspace_mb=4096 >> gpurun_out/r04_slow_phase.jsonl 2>>gpurun_out/sp_err.txt
  timeout 120 python tools/slow_phase_runs.py pretouch=1 >> gpurun_out/r04_slow_phase.jsonl 2>>gpurun_out/sp_err.txt
done
cat gpurun_out/r04_slow_phase.jsonl
