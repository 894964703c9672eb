mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -k "multi_device or two_processes or independent or timeout_is_reported or rccl or exchange" 2>&1 | tail -8 > gpurun_out/r05_run15_tests.txt
cat gpurun_out/r05_run15_tests.txt
