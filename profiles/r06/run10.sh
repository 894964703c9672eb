# round 6, run 10: priority of the units beyond one per wave (1, 2, 3), solving wave at priority 3; c1 with five waves
mkdir -p gpurun_out/r06
timeout 1200 python profiles/knob_ab.py "c2:cold:1:12 c2:steady:1:8 c2:cold:2:16 c2:cold:8:40 c1:cold:1:60" \
  "SAGEICP_LOOP_PRIO=0 SAGEICP_LOOP_DEAL=0" "SAGEICP_LOOP_PRIO=3" "SAGEICP_LOOP_PRIO=2" "SAGEICP_LOOP_PRIO=1" "SAGEICP_LOOP_PRIO=3 SAGEICP_LOOP_DEAL=0" 2>&1 | tee gpurun_out/r06/prio_extra_ab.txt
