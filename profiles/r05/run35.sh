# the frame's sort in one launch: parity with the library sort, then what it buys per frame
(timeout 1500 python -m pytest tests/test_loop_kernel.py -q -x -k "frame_sort" 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -15)
for m in 0 1; do echo "== SAGEICP_SORT_ONE_LAUNCH=$m"; for wl in "c1 cold" "c2 cold"; do SAGEICP_SORT_ONE_LAUNCH=$m timeout 600 python bench.py --workload ${wl% *} --params ${wl#* } --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['workload'][:10], d['value'], d['ms_per_step'], d['roofline'].get('avg_launch_us'))"; done; done
