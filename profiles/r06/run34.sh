mkdir -p gpurun_out/r06
timeout 2400 python -m pytest tests -m gpu -x -q --durations=6 2>&1 | tail -12 > gpurun_out/r06/run34_tests.txt
cat gpurun_out/r06/run34_tests.txt
