timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lanes_per_query_variant" 2>&1 | grep -E "passed|failed|rror|assert" | tail -8
