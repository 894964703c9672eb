# winner's coordinates parked in LDS by the lane that takes a candidate (no second gather in the epilogue):
# default build (74 VGPRs -> 6 waves/SIMD) against the same code held to 7 waves/SIMD (72 VGPRs, 20 B of spills)
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/gputests_run09.txt 2>&1; grep -n "passed\|failed" gpurun_out/gputests_run09.txt; grep -n "Error\|assert" gpurun_out/gputests_run09.txt | head -20
timeout 900 python profiles/knob_probe.py "" "KNOB_LIB=$PWD/sage-icp_amd/_probe/libsageicp_occ7.so" "" "KNOB_LIB=$PWD/sage-icp_amd/_probe/libsageicp_occ7.so" > gpurun_out/hold_lds.txt 2>&1; cat gpurun_out/hold_lds.txt
