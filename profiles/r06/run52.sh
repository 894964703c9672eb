# copies of the one-launch loop's accumulators (8 = one per XCD): contention of 208 workgroups per word against a wider read
mkdir -p gpurun_out/r06
AB_WORKLOADS="c2:cold:12 c1:cold:60 c2:steady:8" timeout 1200 python profiles/ab_probe.py product sage-icp_amd/_probe/libsageicp_rep16.so sage-icp_amd/_probe/libsageicp_rep4.so 2>&1 | tee gpurun_out/r06/replicas_ab.txt
