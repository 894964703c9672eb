mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_reference_order_map.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert|frame" | tail -8
