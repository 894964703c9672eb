# round 6, run 09: the extra units to the workgroups on the CUs that hold the fewest (census), on top of priorities + dealing
mkdir -p gpurun_out/r06
export SAGEICP_LOOP_DEAL=1 SAGEICP_LOOP_PRIO=4
timeout 1200 python profiles/knob_ab.py "c2:cold:1:12 c2:steady:1:8 c2:cold:2:16 c2:cold:4:20 c1:cold:1:60" \
  "SAGEICP_LOOP_CENSUS=0" "SAGEICP_LOOP_CENSUS=1" 2>&1 | tee gpurun_out/r06/census_ab.txt
