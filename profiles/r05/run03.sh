mkdir -p gpurun_out
timeout 600 python profiles/loop_debug2.py > gpurun_out/r05_run03_debug.txt 2>&1
cat gpurun_out/r05_run03_debug.txt
