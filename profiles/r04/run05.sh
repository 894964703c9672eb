# which workgroups of k_loop are the slow ones?
mkdir -p gpurun_out
(timeout 300 python profiles/loop_times.py 8 cold; timeout 300 python profiles/loop_times.py 4 steady) > gpurun_out/r04_loop_times2.txt 2>&1
cat gpurun_out/r04_loop_times2.txt
