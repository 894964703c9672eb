# round 6, run 31: chained k_icp launches (no k_fin, the solving wave resident): c4 / c5 / c2 through k_icp, A/B + poses
mkdir -p gpurun_out/r06
timeout 1500 python profiles/knob_ab.py "c4:steady:1:4 c4:cold:1:4 c5:dense:1:10" "SAGEICP_CHAIN=0" "SAGEICP_CHAIN=1" 2>&1 | tee gpurun_out/r06/chain_ab.txt
SAGEICP_LOOP=0 timeout 600 python profiles/knob_ab.py "c2:cold:1:8 c1:cold:1:30" "SAGEICP_CHAIN=0" "SAGEICP_CHAIN=1" 2>&1 | tee -a gpurun_out/r06/chain_ab.txt
