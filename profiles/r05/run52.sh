(SAGEICP_VARIANT_LIB=sage-icp_amd/_probe/libsageicp_tailsplit.so timeout 2400 python -m pytest tests/test_loop_kernel.py -q -x 2>&1 | grep -E "passed|failed|error|assert" | tail -5)
timeout 2400 python profiles/ab_probe.py product sage-icp_amd/_probe/libsageicp_tailsplit.so 2>&1 | grep "ms/frame" | grep -v c4
