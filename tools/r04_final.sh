#!/bin/bash
export GRANDPLUS_SYNTH_CACHE=/dev/shm/gp_synth
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > gpurun_out/gpu_tests.txt
tail -4 gpurun_out/gpu_tests.txt
PROFILE_ROUND=r04 bash tools/collect_profiles.sh r04c > gpurun_out/r04c_collect.log 2>&1
tail -7 gpurun_out/r04c_collect.log
GRANDPLUS_LIB=libgrandplus_skt.so timeout 600 python tools/sk_phases.py mag 65536 > gpurun_out/r04c/sk_phases.txt 2>&1
GRANDPLUS_LIB=libgrandplus_skt.so timeout 600 python tools/sk_phases.py mag 4096 max_workgroups=8 >> gpurun_out/r04c/sk_phases.txt 2>&1
rm -rf gpurun_out/r04c/trace gpurun_out/r04c/pmc/*/*/*.db 2>/dev/null
GRANDPLUS_STRESS_REPS=30 GRANDPLUS_STRESS_ROWS=65536 timeout 900 python -m pytest tests/test_gpu_stress.py -q -x -k "2-mag or 2-reddit" 2>&1 | tail -3 > gpurun_out/r04c/soak.txt
cat gpurun_out/r04c/soak.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
