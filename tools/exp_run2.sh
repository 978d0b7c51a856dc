tools/instr_split.sh mag 2>&1 | tail -3
for w in pubmed cora amazon2m; do
  for cfg in "1024 163840" "512 81920"; do set -- $cfg
    r=65536; [ $w = amazon2m ] && r=4096
    python bench.py --workload $w --seeds-per-gpu $r --steps 3 --warmup 1 --no-cpu-baseline --block-threads $1 --lds-bytes $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$w $1x$2', round(d['value']), d['roofline']['kernel_ms_avg'])"
  done
done
for w in mag reddit; do for cfg in "1024 163840" "512 81920"; do set -- $cfg
    python bench.py --workload $w --seeds-per-gpu 65536 --steps 3 --warmup 1 --no-cpu-baseline --block-threads $1 --lds-bytes $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$w $1x$2', round(d['value']), d['roofline']['kernel_ms_avg'])"
done; done
