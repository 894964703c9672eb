timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/gputests_run45.txt 2>&1; grep -n "passed\|failed" gpurun_out/gputests_run45.txt | tail -3
