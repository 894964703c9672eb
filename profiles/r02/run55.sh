# full GPU suite + smoke + default bench on the final code of session 9
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -6
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py 2>/dev/null | grep '^{' > gpurun_out/bench_default_s9.json; python -c "
import json; d=json.load(open('gpurun_out/bench_default_s9.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['cpu_baseline']['value'])"
