# the GPU suite three times over on one box (flakiness check before the round ends)
for i in 1 2 3; do timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/gputests_rep$i.txt 2>&1; grep -n "passed\|failed" gpurun_out/gputests_rep$i.txt; done
