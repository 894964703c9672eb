# after the listed-sweep fix: full GPU suite, then the rocprofv3 evidence of all four workloads on this tree
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/gputests_run39.txt 2>&1; grep -n "passed\|failed" gpurun_out/gputests_run39.txt | tail -3
for spec in "c2 cold" "c4 steady" "c1 cold" "c5 dense"; do
  bash profiles/run_profiles.sh r06 $spec > /dev/null 2>&1
done
du -sh gpurun_out/prof_r06_*
