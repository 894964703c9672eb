timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert|Falsifying|seed=" | tail -12
