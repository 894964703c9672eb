# round 6, run 05: first units dealt by SIMD + slot, priorities by unit rank: interleaved A/B
mkdir -p gpurun_out/r06
timeout 1200 python profiles/knob_ab.py "c2:cold:1:12 c2:steady:1:8 c2:cold:2:16 c2:cold:4:20" \
  "SAGEICP_LOOP_PRIO=0" "SAGEICP_LOOP_DEAL=1" "SAGEICP_LOOP_PRIO=4" "SAGEICP_LOOP_PRIO=5" \
  "SAGEICP_LOOP_DEAL=1 SAGEICP_LOOP_PRIO=4" "SAGEICP_LOOP_DEAL=1 SAGEICP_LOOP_PRIO=5" \
  "SAGEICP_LOOP_DEAL=1 SAGEICP_LOOP_PRIO=3 SAGEICP_LOOP_PRIO_LO=1200 SAGEICP_LOOP_PRIO_HI=1800" 2>&1 | tee gpurun_out/r06/deal_ab.txt
