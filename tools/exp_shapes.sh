#!/bin/bash
# Launch-shape sweep of the GFPush kernel on one workload: block threads x LDS per workgroup.
W=${1:-mag}; OUT=${2:-gpurun_out/exp_shapes}; mkdir -p $OUT
run() { # name lib block lds
  GRANDPLUS_LIB=$2 python bench.py --workload $W --seeds-per-gpu 16384 --steps 3 --warmup 1 --no-cpu-baseline --block-threads $3 --lds-bytes $4 2>$OUT/$1.err | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$W $1', round(d['value']), d['roofline']['kernel_ms_avg'], d['detail']['workgroups'], d['detail']['block_threads'], d['detail']['lds_bytes'])"
}
run base1024x160 libgrandplus.so 1024 163840
run b768x160 libgrandplus.so 768 163840
run b512x160_256vgpr libgrandplus_w2.so 512 163840
run b512x80 libgrandplus.so 512 81920
run b256x40 libgrandplus.so 256 40960
run b256x53 libgrandplus.so 256 54272
run base1024x160 libgrandplus.so 1024 163840
