# rank-based (atomic-free) region allocation in the device map update: tests, phase probe, kernel trace A/B
timeout 900 python -m pytest tests/test_map_update_device.py tests/test_pipeline.py -m gpu -x -q > gpurun_out/gputests_run22.txt 2>&1; grep -n "passed\|failed" gpurun_out/gputests_run22.txt; grep -n "Error\|assert" gpurun_out/gputests_run22.txt | head
bash profiles/r03/run21.sh
sed -i 's/^kt main _main A=1$//' profiles/r03/run19.sh
bash profiles/r03/run19.sh
