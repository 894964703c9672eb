# LDS-staged k_up_insert: update tests, then the A/B of run19 (classes on / off)
timeout 900 python -m pytest tests/test_map_update_device.py tests/test_pipeline.py -m gpu -x -q > gpurun_out/gputests_run20.txt 2>&1; grep -n "passed\|failed" gpurun_out/gputests_run20.txt; grep -n "Error\|assert" gpurun_out/gputests_run20.txt | head
sed -i 's/^kt main _main A=1$//' profiles/r03/run19.sh
bash profiles/r03/run19.sh
