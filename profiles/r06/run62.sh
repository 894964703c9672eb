timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "chained_launches_under" 2>&1 | tail -30
