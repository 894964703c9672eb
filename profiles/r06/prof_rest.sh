# round 6: rocprofv3 evidence for c4 steady, c1 cold, c5 dense
for spec in "c4 steady" "c1 cold" "c5 dense"; do
  bash profiles/run_profiles.sh r06 $spec > /dev/null 2>&1
done
du -sh gpurun_out/prof_r06_*
