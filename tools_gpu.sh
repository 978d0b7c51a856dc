mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
run() { GRANDPLUS_DIAG=$1 timeout 900 python bench.py --workload $2 --steps 3 --warmup 1 --seeds-per-gpu 8192 --no-cpu-baseline --block-threads $3 --lds-bytes $4 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$2 bt=$3 lds=$4 diag=$1', d['value'], 'rows/s kernel_ms', d['roofline']['kernel_ms_avg'], 'wgs', d['detail']['workgroups'], d['detail'].get('diag_phase_share'), 'lds/glb levels', d['detail']['lds_levels'], d['detail']['global_levels'])"; }
for w in mag pubmed; do
run 0 $w 1024 163840
run 0 $w 512 163840
run 1 $w 1024 163840
run 1 $w 512 163840
done
