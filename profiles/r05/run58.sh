(SAGEICP_VARIANT_LIB=sage-icp_amd/_probe/libsageicp_w2bounds.so timeout 2400 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | grep -E "passed|failed|error|assert" | tail -4)
timeout 2400 python profiles/ab_probe_big.py product sage-icp_amd/_probe/libsageicp_w2bounds.so 2>&1 | grep "ms/frame"
