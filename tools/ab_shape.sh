#!/bin/bash
# A/B of launch shapes on one box: tools/ab_shape.sh "lib|block|lds ..." [workload ...]   (block 0 = the automatic shape)
CFGS=$1; shift
W=${@:-mag reddit pubmed cora amazon2m}
ROWS=${ROWS:-16384}; STEPS=${STEPS:-3}
for w in $W; do
  r=$ROWS; [ $w = amazon2m ] && r=4096
  for rep in 1 2; do for cfg in $CFGS; do
    IFS='|' read lib blk lds <<< "$cfg"
    GRANDPLUS_LIB=$lib python bench.py --workload $w --seeds-per-gpu $r --steps $STEPS --warmup 1 --no-cpu-baseline --no-host-api --no-next-rows --block-threads $blk --lds-bytes $lds 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$w $cfg', round(d['value']), d['roofline']['kernel_ms_avg'])"
  done; done
done
