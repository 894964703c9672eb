P=sage-icp_amd/_probe
timeout 2400 python profiles/ab_probe.py product $P/libsageicp_poll2.so $P/libsageicp_poll4.so $P/libsageicp_poll16.so 2>&1 | grep "ms/frame" | grep -v "c4"
