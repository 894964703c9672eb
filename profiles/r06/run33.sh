mkdir -p gpurun_out/r06
timeout 2400 python -m pytest tests -m gpu -x -q --durations=15 2>&1 | tail -25 > gpurun_out/r06/run33_tests.txt
cat gpurun_out/r06/run33_tests.txt
