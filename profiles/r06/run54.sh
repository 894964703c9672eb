# final tree of the round: full GPU suite, smoke, the rocprofv3 evidence of all four workloads, the driver's bench command
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/gputests_run54.txt 2>&1; grep -n "passed\|failed" gpurun_out/gputests_run54.txt | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
for spec in "c2 cold" "c4 steady" "c1 cold" "c5 dense"; do
  bash profiles/run_profiles.sh r06 $spec > /dev/null 2>&1
done
du -sh gpurun_out/prof_r06_*
