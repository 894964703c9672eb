# round 5, first run of the rebuilt one-launch loop (groups of queries in LDS, solving wave as its own kernel)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_loop_kernel.py -x -q 2>&1 | tail -15 > gpurun_out/r05_run01_looptests.txt
cat gpurun_out/r05_run01_looptests.txt
timeout 900 python profiles/loop_sweep.py c2 cold 1 3 > gpurun_out/r05_run01_sweep_c2.txt 2>&1
cat gpurun_out/r05_run01_sweep_c2.txt
