# round 6, run 12: two-step packed fp32 filter + cheaper scan bookkeeping against the build before it (same box), then the loop tests
mkdir -p gpurun_out/r06
timeout 900 python profiles/ab_probe.py sage-icp_amd/_probe/libsageicp_prev.so product 2>&1 | tee gpurun_out/r06/filter2_ab.txt
timeout 900 python -m pytest tests/test_loop_kernel.py tests/test_gpu_parity.py -m gpu -x -q -k "lanes or loop or c2 or c5 or property or golden or near" 2>&1 | grep -E "passed|failed|error" | tail -3
