# upper bound of what skipping scans could buy (wrong answers; cost per iteration only; the probe builds never converge: 200 iterations)
P=sage-icp_amd/_probe
SAGEICP_MAX_ITER=200 timeout 1500 python profiles/skin_probe.py $P/libsageicp_skin0.so $P/libsageicp_skin100.so $P/libsageicp_skin90.so $P/libsageicp_skin70.so 2>&1 | grep -v "^$" | tee gpurun_out/r05_skin_probe.txt
