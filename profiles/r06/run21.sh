# round 6, run 21: GPU test suite (env knobs read once + reload, loop status, heaviest-first k_icp, ...)
mkdir -p gpurun_out/r06
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 > gpurun_out/r06/run21_tests.txt
cat gpurun_out/r06/run21_tests.txt
