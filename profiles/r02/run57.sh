python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "independent or two_processes" 2>&1 | grep -E "passed|failed|rror|assert" | tail -6
