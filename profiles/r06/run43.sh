# solving wave: the pose published before the bookkeeping stores — A/B against the library before the change
mkdir -p gpurun_out/r06
AB_WORKLOADS="c2:cold:12 c1:cold:60 c2:steady:8" timeout 900 python profiles/ab_probe.py sage-icp_amd/_probe/libsageicp_prev.so product 2>&1 | tee gpurun_out/r06/publish_first_ab.txt
