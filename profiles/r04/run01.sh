# first run of the one-launch loop (k_loop): its tests, then A/B timing, then the whole GPU suite with it on by default
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_loop_kernel.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r04_loop_tests.txt
cat gpurun_out/r04_loop_tests.txt
timeout 900 python profiles/loop_probe.py > gpurun_out/r04_loop_probe.txt 2>&1
tail -60 gpurun_out/r04_loop_probe.txt
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r04_gputests_run01.txt
cat gpurun_out/r04_gputests_run01.txt
