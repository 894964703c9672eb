# where the frame sort pays: the same frames with the sort replaced by a copy, by frame size
mkdir -p gpurun_out/r06
timeout 1200 python profiles/knob_ab.py "c1:cold:1:60 c1:steady:1:60 c2:cold:16:60 c2:cold:8:40 c2:cold:4:30 c2:cold:2:20 c2:cold:1:10 c2:steady:4:30" "" "SAGEICP_DEBUG_NO_SORT=1" 2>&1 | tee gpurun_out/r06/nosort_ab.txt
