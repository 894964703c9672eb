mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_loop_kernel.py -x -q 2>&1 | tail -5 > gpurun_out/r05_run09_looptests.txt
cat gpurun_out/r05_run09_looptests.txt
SWEEP_LW=2 SWEEP_NW=0,8 SAGEICP_LOOP_DEBUG=1 timeout 900 python profiles/loop_sweep.py c2 cold 1 5 > gpurun_out/r05_run09_sweep_c2.txt 2>&1
grep -v "^sageicp" gpurun_out/r05_run09_sweep_c2.txt; grep "^sageicp" gpurun_out/r05_run09_sweep_c2.txt | sort | uniq -c
timeout 600 python profiles/loop_times.py 1 cold > gpurun_out/r05_run09_times_c2.txt 2>&1
head -14 gpurun_out/r05_run09_times_c2.txt; tail -12 gpurun_out/r05_run09_times_c2.txt
timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/r05_run09_bench.json; python -c "
import json; d=json.load(open('gpurun_out/r05_run09_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['loop_form'])"
