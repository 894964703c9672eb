mkdir -p gpurun_out/r06
AB_WORKLOADS="c2:cold:12" timeout 900 python profiles/ab_probe.py sage-icp_amd/_probe/libsageicp_ad10fe3.so sage-icp_amd/_probe/libsageicp_2e0bcf0.so product 2>&1 | tee gpurun_out/r06/regress_ab.txt
