(timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "counters_can_be" 2>&1 | grep -E "passed|failed|error" | tail -3)
for w in "c2 cold" "c1 cold" "c4 steady" "c5 dense"; do timeout 900 python bench.py --workload ${w% *} --params ${w#* } --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['config']['workload'][:10], d['value'], d['ms_per_step'], r.get('avg_launch_us'), r.get('candidates_per_query'), r.get('pairs_evaluated_frac'))"; done
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['parity'])"
