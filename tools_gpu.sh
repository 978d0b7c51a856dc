timeout 900 python bench.py 2>/dev/null | tail -1
