P=sage-icp_amd/_probe
timeout 1500 python profiles/cap_probe.py $P/libsageicp_cap27.so $P/libsageicp_cap8.so $P/libsageicp_cap4.so $P/libsageicp_cap2.so 2>&1 | grep lanes | tee gpurun_out/r05_cap_probe.txt
