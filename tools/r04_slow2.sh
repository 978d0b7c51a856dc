#!/bin/bash
# Slow-phase experiment, part 2: the round-3 workspace (13.6 GB: every slab sized for 1.5 x the largest level ever seen) against the
# 4.9 GB the same kernel starts with and the 2 GB of the round-4 default, 12 fresh processes each, 68 timed launches per process.
export GRANDPLUS_SYNTH_CACHE=/dev/shm/gp_synth
mkdir -p gpurun_out
: > gpurun_out/r04_slow_phase2.jsonl
for i in $(seq 1 ${RUNS:-12}); do
  timeout 120 python tools/slow_phase_runs.py kernel=1 est_level_edges=143360 >> gpurun_out/r04_slow_phase2.jsonl 2>gpurun_out/sp_err.txt
  timeout 120 python tools/slow_phase_runs.py kernel=1 >> gpurun_out/r04_slow_phase2.jsonl 2>>gpurun_out/sp_err.txt
  timeout 120 python tools/slow_phase_runs.py >> gpurun_out/r04_slow_phase2.jsonl 2>>gpurun_out/sp_err.txt
done
python - <<'PY'
import json
from collections import defaultdict
g=defaultdict(list)
for l in open('gpurun_out/r04_slow_phase2.jsonl'):
    if l.startswith('{'):
        r=json.loads(l); g[json.dumps(r['opts'])].append(r)
for k,v in g.items():
    print(k, 'processes', len(v), 'workspace_gb', v[0]['workspace_gb'], 'median ms', sorted(x['median'] for x in v)[len(v)//2], 'max of max', max(x['max'] for x in v), 'slow launches', sum(x['slow'] for x in v), 'of', sum(x['n'] for x in v))
PY
