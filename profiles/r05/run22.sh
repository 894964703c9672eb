mkdir -p gpurun_out
for div in 2 4 8; do
SWEEP_LW=2,3,4 SWEEP_NW=0,4,8 timeout 900 python profiles/loop_sweep.py c2 cold $div 8 2>&1 | grep -v "^sageicp" | grep -v "contig=1" > gpurun_out/r05_sweep_c2_div$div.txt; cat gpurun_out/r05_sweep_c2_div$div.txt
done
SWEEP_LW=2,3 SWEEP_NW=0 timeout 900 python profiles/loop_sweep.py c4 steady 8 4 2>&1 | grep -v "^sageicp" > gpurun_out/r05_sweep_c4_div8.txt; cat gpurun_out/r05_sweep_c4_div8.txt
SWEEP_LW=3,4 SWEEP_NW=0,4,8 timeout 900 python profiles/loop_sweep.py c1 cold 1 20 2>&1 | grep -v "^sageicp" | grep -v "contig=1" > gpurun_out/r05_sweep_c1.txt; cat gpurun_out/r05_sweep_c1.txt
