# k_loop with XCD stripes and the row shift on voxel crossings: tests, timeline, A/B, whole suite
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_loop_kernel.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r04_loop_tests.txt
cat gpurun_out/r04_loop_tests.txt
(timeout 300 python profiles/loop_times.py 8 cold; timeout 300 python profiles/loop_times.py 4 steady) > gpurun_out/r04_loop_times3.txt 2>&1
grep -v "^   [ 0-9]* |" gpurun_out/r04_loop_times3.txt
timeout 900 python profiles/loop_probe.py quick > gpurun_out/r04_loop_probe.txt 2>&1
cat gpurun_out/r04_loop_probe.txt
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/r04_gputests_run06.txt
cat gpurun_out/r04_gputests_run06.txt
