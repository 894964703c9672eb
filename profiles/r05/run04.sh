mkdir -p gpurun_out
timeout 300 python profiles/loop_debug.py c2 0.05 cold 2>&1 | grep -v "^sageicp:" > gpurun_out/r05_run04_debug.txt
head -40 gpurun_out/r05_run04_debug.txt
timeout 1500 python -m pytest tests/test_loop_kernel.py -x -q 2>&1 | tail -15 > gpurun_out/r05_run04_looptests.txt
cat gpurun_out/r05_run04_looptests.txt
timeout 900 python profiles/loop_sweep.py c2 cold 1 3 > gpurun_out/r05_run04_sweep_c2.txt 2>&1
cat gpurun_out/r05_run04_sweep_c2.txt
