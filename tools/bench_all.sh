#!/bin/bash
# Runs bench.py on the five BASELINE configurations (no CPU leg) and prints workload, rows/s and kernel ms.
for w in mag reddit pubmed cora; do
  python bench.py --workload $w --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$w', round(d['value']), d['roofline']['kernel_ms_avg'], d['roofline']['frac'], d['detail']['degree_lookups_per_row'])"
done
python bench.py --workload amazon2m --seeds-per-gpu 4096 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('amazon2m', round(d['value']), d['roofline']['kernel_ms_avg'], d['roofline']['frac'], d['detail']['degree_lookups_per_row'])"
