(timeout 1500 python -m pytest tests/test_loop_kernel.py -q -x -k "frame_sort" 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -15)
bash profiles/r05/run36.sh
