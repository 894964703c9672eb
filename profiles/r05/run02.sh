mkdir -p gpurun_out
timeout 600 python profiles/loop_debug.py c2 0.05 cold > gpurun_out/r05_run02_debug.txt 2>&1
head -120 gpurun_out/r05_run02_debug.txt
SAGEICP_LOOP_DEBUG=1 SAGEICP_LOOP=2 timeout 300 python profiles/loop_sweep.py c2 cold 1 1 2>&1 | grep -m3 "sageicp:" 
