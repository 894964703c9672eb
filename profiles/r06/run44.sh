# the tree with small frames unsorted + publish-first: full GPU suite, then the rocprofv3 evidence of all four workloads
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/gputests_run44.txt 2>&1; grep -n "passed\|failed" gpurun_out/gputests_run44.txt | tail -3
for spec in "c2 cold" "c4 steady" "c1 cold" "c5 dense"; do
  bash profiles/run_profiles.sh r06 $spec > /dev/null 2>&1
done
du -sh gpurun_out/prof_r06_*
