mkdir -p gpurun_out/r06
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 > gpurun_out/r06/run32_tests.txt
cat gpurun_out/r06/run32_tests.txt
