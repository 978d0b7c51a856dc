#!/bin/bash
# Everything the round's profiles/ directory is built from, in one GPU-box call:
#   tools/collect_profiles.sh <tag>      -> gpurun_out/<tag>/{kernel_stats.csv, pmc_summary.json, bench_lines.json, ...}
# Every rocprofv3 pass runs under its own timeout (a failed pass can hang while finalizing).
TAG=${1:-final}; OUT=gpurun_out/$TAG
export TMPDIR=/tmp
export GRANDPLUS_SYNTH_CACHE=${GRANDPLUS_SYNTH_CACHE:-/dev/shm/gp_synth}
mkdir -p $OUT
# 1. kernel trace of the default bench line (MAG shape)
timeout -k 5 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --prewarm 0 --steps 5 --warmup 2 --no-cpu-baseline --no-host-api --no-next-rows --no-cold-call --opt measure_choice=0 > $OUT/trace.log 2>&1
cp $(ls -t $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
# 2. PMC passes: all counter groups for the headline workload, the memory-side and instruction groups for the other four
#    (every bench line then carries counter traffic -- VERDICT r4 #4)
tools/collect_pmc.sh mag $OUT/pmc_mag
python tools/pmc_summary.py $OUT/pmc_mag $OUT/mag_pmc_summary.json --workload mag > /dev/null
for w in reddit pubmed cora; do
  PMC_GROUPS=traffic tools/collect_pmc.sh $w $OUT/pmc_$w
  python tools/pmc_summary.py $OUT/pmc_$w $OUT/${w}_pmc_summary.json --workload $w > /dev/null
done
PMC_GROUPS=traffic tools/collect_pmc.sh amazon2m $OUT/pmc_amazon2m --seeds-per-gpu 12350
python tools/pmc_summary.py $OUT/pmc_amazon2m $OUT/amazon2m_pmc_summary.json --workload amazon2m --rows 12350 > /dev/null
# (bench.py reads roofline.traffic from profiles/: put this run's summaries there before the bench lines are taken)
if [ -n "$PROFILE_ROUND" ]; then for w in mag reddit pubmed cora amazon2m; do cp $OUT/${w}_pmc_summary.json profiles/${PROFILE_ROUND}_${w}_pmc_summary.json; done; fi
rm -rf $OUT/pmc_*/g*/                      # (the raw counter dumps: tens of MB)
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
