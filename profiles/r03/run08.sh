# winner's coordinates kept from the scan (no gather in the epilogue): suite subset + c2 / c4 lines
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/gputests_run08.txt 2>&1; grep -n "passed\|failed" gpurun_out/gputests_run08.txt; grep -n "Error\|assert" gpurun_out/gputests_run08.txt | head -20
for wl in c2 c4; do
    timeout 600 python bench.py --workload $wl --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$wl', d['value'], d['ms_per_step'], d['config']['iterations_per_frame'], d['roofline']['avg_launch_us'])"
done | tee gpurun_out/track_ab.txt
