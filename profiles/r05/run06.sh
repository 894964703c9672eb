mkdir -p gpurun_out
timeout 600 python profiles/loop_times.py 1 cold > gpurun_out/r05_run06_times_c2.txt 2>&1
cat gpurun_out/r05_run06_times_c2.txt
