# round 6, run 08: with priorities + dealing, how many workgroups? (6 per CU exactly = 1536)
mkdir -p gpurun_out/r06
export SAGEICP_LOOP_DEAL=1 SAGEICP_LOOP_PRIO=4
timeout 1200 python profiles/knob_ab.py "c2:cold:1:12 c2:steady:1:8" \
  "SAGEICP_LOOP_MAX_WGS=1664" "SAGEICP_LOOP_MAX_WGS=1280" "SAGEICP_LOOP_MAX_WGS=1408" "SAGEICP_LOOP_MAX_WGS=1536" "SAGEICP_LOOP_MAX_WGS=1600" "SAGEICP_LOOP_MAX_WGS=1696" 2>&1 | tee gpurun_out/r06/wgs_ab.txt
timeout 600 python profiles/knob_ab.py "c1:cold:1:60" "SAGEICP_LOOP_WAVES=4" "SAGEICP_LOOP_WAVES=5" "SAGEICP_LOOP_WAVES=8" "SAGEICP_LOOP_WAVES=6" "SAGEICP_LOOP_WAVES=2"  2>&1 | tee gpurun_out/r06/c1_waves_ab.txt
