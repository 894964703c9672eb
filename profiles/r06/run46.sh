# waves per workgroup of k_icp (frames beyond the LDS): a workgroup's slots are free again only when its slowest wave ends
mkdir -p gpurun_out/r06
AB_WORKLOADS="c4:steady:4 c5:dense:10 c4:cold:4" timeout 1500 python profiles/ab_probe.py product sage-icp_amd/_probe/libsageicp_w2s16.so sage-icp_amd/_probe/libsageicp_w2s8.so sage-icp_amd/_probe/libsageicp_w8s4.so 2>&1 | tee gpurun_out/r06/icp_waves_ab.txt
