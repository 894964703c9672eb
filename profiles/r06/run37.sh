mkdir -p gpurun_out/r06
AB_WORKLOADS="c2:cold:12 c1:cold:60" timeout 900 python profiles/ab_probe.py sage-icp_amd/_probe/libsageicp_2e0bcf0.so product 2>&1 | tee gpurun_out/r06/regress_ab2.txt
timeout 900 python profiles/knob_ab.py "c4:steady:1:4 c5:dense:1:10" "SAGEICP_CHAIN=0" "SAGEICP_CHAIN=1" 2>&1 | grep -v "^c" 
