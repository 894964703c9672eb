timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "property" 2>&1 | grep -E "passed|failed|rror|assert|Falsifying|seed=|^E " | tail -25
