#!/bin/bash
# Option sweep of the sketch kernel on ONE box: tools/opt_sweep.sh <workload> "<key=value ...>" "<key=value ...>" ...
# (each quoted group is one configuration of gp_set_option keys; the empty group "" is the default)
W=${1:-mag}; shift
export GRANDPLUS_SYNTH_CACHE=${GRANDPLUS_SYNTH_CACHE:-/dev/shm/gp_synth}
export SKQ_ONLY=${SKQ_ONLY:-"sketch 768"}
mkdir -p gpurun_out $GRANDPLUS_SYNTH_CACHE
: > gpurun_out/opt_sweep.txt
for cfg in "$@"; do
  echo -n "[$cfg] " >> gpurun_out/opt_sweep.txt
  timeout 300 python tools/sk_quick.py $W 65536 $cfg 2>&1 | grep " best " | cut -c1-140 >> gpurun_out/opt_sweep.txt
done
cat gpurun_out/opt_sweep.txt
