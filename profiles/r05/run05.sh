mkdir -p gpurun_out
SAGEICP_LOOP_DEBUG=1 timeout 900 python profiles/loop_sweep.py c2 cold 1 3 > gpurun_out/r05_run05_sweep_c2.txt 2>&1
grep -v "^sageicp" gpurun_out/r05_run05_sweep_c2.txt; grep "^sageicp" gpurun_out/r05_run05_sweep_c2.txt | sort | uniq -c
