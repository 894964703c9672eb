# bound tests + voxel-index divisions split over the lanes of a query, per-wave counters through DPP + fire-and-forget atomics
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/gputests_run13.txt 2>&1; grep -n "passed\|failed" gpurun_out/gputests_run13.txt; grep -n "Error\|assert" gpurun_out/gputests_run13.txt | head
for a in "c2 cold" "c4 steady" "c1 cold" "c5 dense"; do set -- $a
  timeout 600 python bench.py --workload $1 --params $2 --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 $2:', d['value'], d['ms_per_step'], d['config']['iterations_per_frame'], d['roofline']['avg_launch_us'])"
done | tee gpurun_out/lanesplit.txt
timeout 600 python profiles/knob_probe.py "" | tee -a gpurun_out/lanesplit.txt
