# the shipped binaries once more: smoke, the driver's bench command, the whole GPU suite
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_driver.json 2> gpurun_out/r05_bench_driver.err; python -c "
import json
d=json.loads(open('gpurun_out/r05_bench_driver.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], r['avg_launch_us'], 'frac', r['frac'], r['counters_fresh'], 'parity', d['parity']['ok'], 'cpu', d['cpu_baseline']['value'])"
(timeout 3000 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -3) | tee gpurun_out/r05_gputests.txt
