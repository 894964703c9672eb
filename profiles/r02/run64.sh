# final state of session 9: GPU suite + smoke + the driver's bench command
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -4
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep '^{' > gpurun_out/bench_final_s9.json; python -c "
import json; d=json.load(open('gpurun_out/bench_final_s9.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
