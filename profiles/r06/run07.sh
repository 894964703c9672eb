# round 6, run 07: units of a workgroup from all over the frame (transpose) x priorities x dealing
mkdir -p gpurun_out/r06
timeout 1200 python profiles/knob_ab.py "c2:cold:1:12 c2:steady:1:8 c2:cold:2:16 c1:cold:1:60" \
  "SAGEICP_LOOP_PRIO=0" "SAGEICP_LOOP_DEAL=1 SAGEICP_LOOP_PRIO=4" "SAGEICP_LOOP_TRANSPOSE=1" "SAGEICP_LOOP_TRANSPOSE=1 SAGEICP_LOOP_PRIO=4" \
  "SAGEICP_LOOP_TRANSPOSE=1 SAGEICP_LOOP_DEAL=1 SAGEICP_LOOP_PRIO=4" 2>&1 | tee gpurun_out/r06/transpose_ab.txt
