# round 6, run 14: two looks in flight (pose granules + abort word in one load; counts likewise) against the build before
mkdir -p gpurun_out/r06
AB_WORKLOADS="c2:cold:20 c1:cold:100 c2:steady:12" timeout 900 python profiles/ab_probe.py sage-icp_amd/_probe/libsageicp_prev.so product 2>&1 | grep -v "c4" | tee gpurun_out/r06/poll2_ab.txt
