mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_loop_kernel.py -x -q 2>&1 | tail -8 > gpurun_out/r05_run14_looptests.txt
cat gpurun_out/r05_run14_looptests.txt
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r05_run14_gputests.txt
cat gpurun_out/r05_run14_gputests.txt
