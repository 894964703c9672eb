mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_loop_kernel.py -x -q 2>&1 | tail -8 > gpurun_out/r05_run13_looptests.txt
cat gpurun_out/r05_run13_looptests.txt
SWEEP_LW=2 SWEEP_NW=0 timeout 900 python profiles/loop_sweep.py c2 cold 1 5 > gpurun_out/r05_run13_sweep_c2.txt 2>&1
cat gpurun_out/r05_run13_sweep_c2.txt
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r05_run13_gputests.txt
cat gpurun_out/r05_run13_gputests.txt
