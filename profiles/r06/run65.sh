# k_icp: the home voxel of a seeded query scanned with its neighbours (as k_loop does) instead of in a pass of its own
mkdir -p gpurun_out/r06
AB_WORKLOADS="c4:steady:4 c5:dense:10 c4:cold:4" timeout 1500 python profiles/ab_probe.py sage-icp_amd/_probe/libsageicp_prev.so sage-icp_amd/_probe/libsageicp_merged.so 2>&1 | tee gpurun_out/r06/merged_home_ab.txt
