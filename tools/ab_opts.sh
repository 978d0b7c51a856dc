#!/bin/bash
export GRANDPLUS_SYNTH_CACHE=${GRANDPLUS_SYNTH_CACHE:-/dev/shm/gp_synth}
# A/B of (library, bench options) pairs on the SAME GPU box:  tools/ab_opts.sh "lib.so|--block-threads 1024 --lds-bytes 81920" "lib2.so|" -- mag reddit
# ROWS=65536 STEPS=3 by default.
PAIRS=(); while [ "$1" != "--" ] && [ -n "$1" ]; do PAIRS+=("$1"); shift; done; shift
W=${@:-mag}
ROWS=${ROWS:-65536}; STEPS=${STEPS:-3}
for w in $W; do
  r=$ROWS; [ $w = amazon2m ] && r=12350
  for rep in 1 2; do for pr in "${PAIRS[@]}"; do
    lib=${pr%%|*}; opts=${pr#*|}
    GRANDPLUS_LIB=$lib timeout 300 python bench.py --workload $w --seeds-per-gpu $r --steps $STEPS --warmup 2 --no-cpu-baseline --no-host-api --no-next-rows $opts 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$w $lib [$opts]', round(d['value']), d['roofline']['kernel_ms_avg'])"
  done; done
done
