# the round-1..3 association of the 3-term norms (libsageicp_hip.v0.so + the oracle built the same way): the parity suites once more
(SAGE_SQNORM3_ORDER=0 timeout 3000 python -m pytest tests/test_gpu_parity.py tests/test_loop_kernel.py -q -x 2>&1 | grep -E "passed|failed|error" | tail -3) | tee gpurun_out/r05_gputests_sqnorm3_order0.txt
