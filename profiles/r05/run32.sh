# final tree of the session so far: whole GPU suite, smoke, the c2-cold profile set, the driver's bench line
(timeout 3000 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -3) > gpurun_out/r05_gputests.txt
cat gpurun_out/r05_gputests.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash profiles/run_profiles.sh r05 c2 cold > /dev/null 2>&1
tail -30 gpurun_out/prof_r05_c2-cold/summary.md
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_driver.json 2> gpurun_out/r05_bench_driver.err; cat gpurun_out/r05_bench_driver.json
