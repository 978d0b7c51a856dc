#!/bin/bash
# Everything a round's profiles/ directory and its soak / fuzz records are built from, in one GPU-box call:
#   PROFILE_ROUND=r05 tools/final_collection.sh <tag>      -> gpurun_out/<tag>/...  (copy what is to be judged into profiles/)
# Needs the -DGP_SK_TIMING build (GRANDPLUS_BUILD_DIAG=1 python -c 'import __graft_entry__ as g; g.build()').
TAG=${1:-final}
export GRANDPLUS_SYNTH_CACHE=/dev/shm/gp_synth
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
mkdir -p gpurun_out/$TAG $GRANDPLUS_SYNTH_CACHE
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > gpurun_out/$TAG/gpu_tests.txt
tail -4 gpurun_out/$TAG/gpu_tests.txt
bash tools/collect_profiles.sh $TAG > gpurun_out/$TAG/collect.log 2>&1
tail -7 gpurun_out/$TAG/collect.log
GRANDPLUS_LIB=libgrandplus_skt.so timeout 600 python tools/sk_phases.py mag 65536 > gpurun_out/$TAG/sk_phases.txt 2>&1
GRANDPLUS_LIB=libgrandplus_skt.so timeout 600 python tools/sk_phases.py mag 4096 max_workgroups=8 >> gpurun_out/$TAG/sk_phases.txt 2>&1
GRANDPLUS_DIAG=1 timeout 600 python tools/level_breakdown.py amazon2m > gpurun_out/$TAG/amazon2m_levels.txt 2>&1
rm -rf gpurun_out/$TAG/trace 2>/dev/null
GRANDPLUS_STRESS_REPS=30 GRANDPLUS_STRESS_ROWS=65536 timeout 900 python -m pytest tests/test_gpu_stress.py -q -x -k "2-mag or 2-reddit" 2>&1 | tail -3 > gpurun_out/$TAG/soak.txt
cat gpurun_out/$TAG/soak.txt
timeout 500 python tools/fuzz_sketch.py 400 > gpurun_out/$TAG/fuzz.txt 2>&1; tail -3 gpurun_out/$TAG/fuzz.txt
FUZZ_KERNEL=1 timeout 300 python tools/fuzz_sketch.py 200 >> gpurun_out/$TAG/fuzz.txt 2>&1; tail -2 gpurun_out/$TAG/fuzz.txt
timeout 900 python tools/slow_propagate_runs.py ${SLOW_PROP_RUNS:-8} mag > gpurun_out/$TAG/slow_propagate.jsonl 2>&1; tail -1 gpurun_out/$TAG/slow_propagate.jsonl
python tools/host_api_latency.py 2>&1 | grep -v amdgpu.ids > gpurun_out/$TAG/host_api_latency.txt; tail -2 gpurun_out/$TAG/host_api_latency.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
