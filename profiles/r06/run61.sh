# waves per workgroup of the one-launch loop on the final kernels (priorities + dealing changed who waits for whom)
mkdir -p gpurun_out/r06
timeout 1500 python profiles/knob_ab.py "c2:cold:1:10 c2:cold:2:16 c1:cold:1:60 c2:cold:8:40" "" "SAGEICP_LOOP_WAVES=5" "SAGEICP_LOOP_WAVES=6" "SAGEICP_LOOP_WAVES=7" "SAGEICP_LOOP_WAVES=8" "SAGEICP_LOOP_WAVES=3" 2>&1 | tee gpurun_out/r06/loop_waves_ab.txt
