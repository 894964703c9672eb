timeout 1800 python profiles/nosort_probe2.py sage-icp_amd/_probe/libsageicp_nosort.so 2>&1 | grep "queries:"
