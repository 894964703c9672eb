# DPP exchanges as one instruction per dword (bound_ctrl, no old operand): parity, then the workloads
(timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_loop_kernel.py -q -x 2>&1 | grep -E "passed|failed|error" | tail -3)
for w in "c2 cold" "c1 cold" "c4 steady" "c5 dense"; do timeout 900 python bench.py --workload ${w% *} --params ${w#* } --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['workload'][:10], d['value'], d['ms_per_step'], d['roofline'].get('avg_launch_us'))"; done
