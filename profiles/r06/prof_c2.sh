# round 6: rocprofv3 evidence, c2 cold (kernel trace + 5 counter passes + the un-profiled line with cpu_baseline and parity)
bash profiles/run_profiles.sh r06 c2 cold > /dev/null 2>&1
tail -40 gpurun_out/prof_r06_c2-cold/summary.md
du -sh gpurun_out/prof_r06_c2-cold
