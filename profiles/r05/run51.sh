timeout 2400 python profiles/ab_probe.py product sage-icp_amd/_probe/libsageicp_flaglds.so 2>&1 | grep "ms/frame"
