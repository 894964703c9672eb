# k_loop: tests, per-iteration timeline (instrumented build), the whole GPU suite with k_loop on by default
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_loop_kernel.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r04_loop_tests.txt
cat gpurun_out/r04_loop_tests.txt
(timeout 300 python profiles/loop_times.py 8 cold; timeout 300 python profiles/loop_times.py 4 steady) > gpurun_out/r04_loop_times.txt 2>&1
cat gpurun_out/r04_loop_times.txt
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r04_gputests_run02.txt
cat gpurun_out/r04_gputests_run02.txt
