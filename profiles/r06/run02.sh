# round 6, run 02: wave priorities in the one-launch loop (s_setprio), same-process interleaved A/B
mkdir -p gpurun_out/r06
timeout 1200 python profiles/knob_ab.py "c2:cold:1:12 c2:steady:1:8 c1:cold:1:60 c2:cold:2:16 c2:cold:8:40" \
  "SAGEICP_LOOP_PRIO=0" "SAGEICP_LOOP_PRIO=1" "SAGEICP_LOOP_PRIO=2" \
  "SAGEICP_LOOP_PRIO=3 SAGEICP_LOOP_PRIO_LO=900 SAGEICP_LOOP_PRIO_HI=1300" \
  "SAGEICP_LOOP_PRIO=3 SAGEICP_LOOP_PRIO_LO=700 SAGEICP_LOOP_PRIO_HI=1000" \
  "SAGEICP_LOOP_PRIO=3 SAGEICP_LOOP_PRIO_LO=1200 SAGEICP_LOOP_PRIO_HI=1800" 2>&1 | tee gpurun_out/r06/prio_ab.txt
