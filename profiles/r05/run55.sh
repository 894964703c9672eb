P=sage-icp_amd/_probe
timeout 2400 python profiles/ab_probe.py product $P/libsageicp_poll16.so $P/libsageicp_poll24.so $P/libsageicp_poll32.so $P/libsageicp_poll48.so 2>&1 | grep "ms/frame" | grep "c2"
