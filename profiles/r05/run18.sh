mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_loop_kernel.py -x -q 2>&1 | tail -4 > gpurun_out/r05_run18_looptests.txt
cat gpurun_out/r05_run18_looptests.txt
SWEEP_LW=2 SWEEP_NW=0 timeout 900 python profiles/loop_sweep.py c2 cold 1 5 2>&1 | grep -v "^sageicp" > gpurun_out/r05_run18_sweep_c2.txt; cat gpurun_out/r05_run18_sweep_c2.txt
timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/r05_run18_bench.json; python -c "
import json; d=json.load(open('gpurun_out/r05_run18_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['loop_form'])"
for wl in "c1 cold" "c4 steady" "c5 dense"; do set -- $wl; timeout 900 python bench.py --workload $1 --params $2 --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/r05_run18_bench_$1.json; python -c "
import json; d=json.load(open('gpurun_out/r05_run18_bench_$1.json')); print('$1', d['value'], d['ms_per_step'], d['config']['iterations_per_frame'], d['roofline']['loop_form'], d['roofline']['lanes_per_query'])"; done
