mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_loop_kernel.py -x -q 2>&1 | tail -5 > gpurun_out/r05_run07_looptests.txt
cat gpurun_out/r05_run07_looptests.txt
SWEEP_LW=2 SWEEP_NW=0,4,8 SAGEICP_LOOP_DEBUG=1 timeout 900 python profiles/loop_sweep.py c2 cold 1 5 > gpurun_out/r05_run07_sweep_c2.txt 2>&1
grep -v "^sageicp" gpurun_out/r05_run07_sweep_c2.txt; grep "^sageicp" gpurun_out/r05_run07_sweep_c2.txt | sort | uniq -c
timeout 600 python profiles/loop_times.py 1 cold > gpurun_out/r05_run07_times_c2.txt 2>&1
head -14 gpurun_out/r05_run07_times_c2.txt; tail -12 gpurun_out/r05_run07_times_c2.txt
