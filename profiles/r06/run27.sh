# round 6, run 27: compile-time variants of the loop under the new priorities: flat order at 4 lanes, poll back-off, scan depth
mkdir -p gpurun_out/r06
AB_WORKLOADS="c2:cold:16 c2:steady:10" timeout 1500 python profiles/ab_probe.py product sage-icp_amd/_probe/libsageicp_flat4.so sage-icp_amd/_probe/libsageicp_poll16.so sage-icp_amd/_probe/libsageicp_poll4.so sage-icp_amd/_probe/libsageicp_depth1.so sage-icp_amd/_probe/libsageicp_depth3.so 2>&1 | tee gpurun_out/r06/variants_ab.txt
