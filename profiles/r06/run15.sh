# round 6, run 15: k_icp on c4: phases of a wave, occupancy timeline of one launch
mkdir -p gpurun_out/r06
timeout 900 python profiles/icp_tail.py c4 steady 20 2>&1 | tee gpurun_out/r06/icp_tail_c4.txt
