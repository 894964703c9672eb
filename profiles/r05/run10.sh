mkdir -p gpurun_out
LOOP_LIB=sage-icp_amd/_probe/libsageicp_occ8.so SWEEP_LW=2 SWEEP_NW=0 SAGEICP_LOOP_DEBUG=1 timeout 900 python profiles/loop_sweep.py c2 cold 1 5 > gpurun_out/r05_run10_sweep_c2_occ8.txt 2>&1
grep -v "^sageicp" gpurun_out/r05_run10_sweep_c2_occ8.txt; grep "^sageicp" gpurun_out/r05_run10_sweep_c2_occ8.txt | sort | uniq -c
