run() { GRANDPLUS_DIAG=$1 timeout 900 python bench.py --workload $2 --steps 3 --warmup 1 --seeds-per-gpu 8192 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$2 diag=$1', d['value'], 'rows/s', 'kernel', d['roofline']['kernel_ms_avg'], d['detail'].get('diag_phase_share'))"; }
for w in mag reddit pubmed cora; do run 0 $w; run 1 $w; done
