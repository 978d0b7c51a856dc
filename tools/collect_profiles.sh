#!/bin/bash
# Everything the round's profiles/ directory is built from, in one GPU-box call:
#   tools/collect_profiles.sh <tag>      -> gpurun_out/<tag>/{kernel_stats.csv, pmc_summary.json, bench_lines.json, ...}
# Every rocprofv3 pass runs under its own timeout (a failed pass can hang while finalizing).
TAG=${1:-final}; OUT=gpurun_out/$TAG
export TMPDIR=/tmp
export GRANDPLUS_SYNTH_CACHE=${GRANDPLUS_SYNTH_CACHE:-/dev/shm/gp_synth}
mkdir -p $OUT
# 1. kernel trace of the default bench line (MAG shape)
timeout -k 5 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --prewarm 0 --steps 5 --warmup 2 --no-cpu-baseline --no-host-api --no-next-rows > $OUT/trace.log 2>&1
cp $(ls -t $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
# 2. PMC passes
tools/collect_pmc.sh mag $OUT/pmc
python tools/pmc_summary.py $OUT/pmc $OUT/pmc_summary.json > /dev/null
# (bench.py reads roofline.traffic from profiles/: put this run's summary there before the bench lines are taken)
[ -n "$PROFILE_ROUND" ] && cp $OUT/pmc_summary.json profiles/${PROFILE_ROUND}_mag_pmc_summary.json
# 3. the bench lines of the five BASELINE configurations, CPU baseline included
python - <<PY
import json, subprocess, sys
out = {}
for w, extra in (("mag", []), ("pubmed", []), ("reddit", []), ("cora", []), ("amazon2m", ["--seeds-per-gpu", "12350"])):
    r = subprocess.run([sys.executable, "bench.py", "--workload", w] + extra, capture_output=True, text=True)
    try:
        out[w] = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:
        out[w] = {"error": str(e), "stderr": r.stderr[-2000:]}
    print(w, out[w].get("value"), flush=True)
json.dump(out, open("$OUT/bench_lines.json", "w"), indent=1)
PY
