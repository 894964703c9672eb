# round 6, run 18: stripe order by the max of a stripe's waves instead of the sum
mkdir -p gpurun_out/r06
timeout 1500 python profiles/knob_ab.py "c4:steady:1:4 c5:dense:1:10" "SAGEICP_LPT=0" "SAGEICP_LPT=1" "SAGEICP_LPT=1 SAGEICP_LPT_MAX=1" 2>&1 | tee gpurun_out/r06/lpt_max_ab.txt
SAGEICP_LOOP=0 timeout 600 python profiles/knob_ab.py "c2:cold:1:8" "SAGEICP_LPT=0" "SAGEICP_LPT=1" "SAGEICP_LPT=1 SAGEICP_LPT_MAX=1" 2>&1 | tee -a gpurun_out/r06/lpt_max_ab.txt
