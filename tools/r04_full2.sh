#!/bin/bash
export GRANDPLUS_SYNTH_CACHE=/dev/shm/gp_synth
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > gpurun_out/gpu_tests.txt
tail -8 gpurun_out/gpu_tests.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_mag.json 2> gpurun_out/bench_mag.err
tail -3 gpurun_out/bench_mag.err
python - <<'PY'
import json
l=json.loads(open('gpurun_out/bench_mag.json').read().strip().splitlines()[-1])
print({k:l[k] for k in ('value','ms_per_step','warmup_effective','anomaly','prewarm_step_ms_median')})
print(l['roofline']); print(l.get('host_api')); print(l['detail']); print(l.get('cpu_baseline'))
PY
