# round 6, run 13: one contiguous eighth of the frame per XCD (equal counts) under the new priorities
mkdir -p gpurun_out/r06
timeout 1200 python profiles/knob_ab.py "c2:cold:1:12 c2:steady:1:8 c2:cold:2:16" \
  "SAGEICP_LOOP_CONTIGUOUS=0" "SAGEICP_LOOP_CONTIGUOUS=1" 2>&1 | tee gpurun_out/r06/contig_ab.txt
