timeout 1800 python profiles/nosort_probe.py sage-icp_amd/_probe/libsageicp_nosort.so 2>&1 | grep "ms/frame"
