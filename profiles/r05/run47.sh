# three lanes per query: correctness first, then the headline frame
(timeout 1800 python -m pytest tests/test_loop_kernel.py -q -x -k "every_shape or several_groups or bits_do_not" 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -12)
for t in 1 0; do echo "== SAGEICP_LOOP_TRI=$t"; SAGEICP_LOOP_TRI=$t SAGEICP_LOOP_DEBUG=1 timeout 600 python bench.py --workload c2 --params cold --no-cpu-baseline 2>&1 | python -c "
import json,sys
L=sys.stdin.read().strip().splitlines()
print([l for l in L if 'one-launch loop for' in l][:1])
d=json.loads(L[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], r.get('avg_launch_us'), r.get('lanes_per_query'), d['config']['iterations_per_frame'], d['config']['pose_error_vs_planted'])"; done
