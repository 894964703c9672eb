mkdir -p gpurun_out
timeout 600 python profiles/loop_times.py 1 cold > gpurun_out/r05_run11_times_c2.txt 2>&1
sed -n 10,40p gpurun_out/r05_run11_times_c2.txt
