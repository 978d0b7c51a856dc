timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "full_scale" 2>&1 | tail -6
