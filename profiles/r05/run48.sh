(timeout 1800 python -m pytest tests/test_loop_kernel.py -q -x -k "every_shape and 5" 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5)
for t in 1; do echo "== SAGEICP_LOOP_TRI=$t"; SAGEICP_LOOP_TRI=$t timeout 600 python bench.py --workload c2 --params cold --no-cpu-baseline 2>&1 | python -c "
import json,sys
L=sys.stdin.read().strip().splitlines()
d=json.loads(L[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], r.get('avg_launch_us'), r.get('lanes_per_query'), d['config']['iterations_per_frame'])"; done
LOOP_LIB=sage-icp_amd/_probe/libsageicp_looptiming.so timeout 600 python profiles/loop_times.py 1 cold c2 2>&1 | grep -E "pose, query|row rebuild|seed,|bounds|scan|argmin|answer|pair terms|body|closing|waiting|queries,"
