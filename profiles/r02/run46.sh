# final check of the round: whole GPU suite, smoke, default bench line (with the refreshed counters), c1 with 4 lanes per query
(timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|rror" | tail -3)
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -3
python bench.py > gpurun_out/final_bench_default.json 2> gpurun_out/final_bench_default.err; python -c "
import json; d=json.load(open('gpurun_out/final_bench_default.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], d['ms_per_step_host_entry'], r['avg_launch_us'], r['frac'], r['hbm_frac'], r['valu_frac'], r['useful_inst_frac'], r['compulsory_gbs'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
python bench.py --workload c1 --no-cpu-baseline > gpurun_out/final_bench_c1.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/final_bench_c1.json')); print('c1', d['value'], d['ms_per_step'], d['roofline']['lanes_per_query'])"
