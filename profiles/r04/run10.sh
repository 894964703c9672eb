# k_skip with 64 queries per workgroup: tests, A/B against k_icp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_skip_kernel.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r04_skip_tests.txt
cat gpurun_out/r04_skip_tests.txt
timeout 1500 python profiles/skip_ab.py c2:cold c2:steady c5:dense > gpurun_out/r04_skip_ab.txt 2>&1
cat gpurun_out/r04_skip_ab.txt
