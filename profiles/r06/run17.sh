# round 6, run 17: k_icp on c4 with the stripes dispatched heaviest first: occupancy timeline of one launch
mkdir -p gpurun_out/r06
timeout 900 python profiles/icp_tail.py c4 steady 20 2>&1 | tee gpurun_out/r06/icp_tail_c4_lpt.txt
