# fixed-point accumulators instead of per-workgroup partials: suite, A/B on c2 / c4 / c1 / c5, k_fin phase stamps
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gputests_run06.txt 2>&1; grep -n "passed\|failed" gpurun_out/gputests_run06.txt; grep -n "Error\|assert" gpurun_out/gputests_run06.txt | head -20
for wl in c2 c4 c1 c5; do
  for p in 0 1; do
    SAGEICP_PARTIALS=$p timeout 600 python bench.py --workload $wl --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$wl partials=$p', d['value'], d['ms_per_step'], d['config']['iterations_per_frame'], d['roofline']['avg_launch_us'])"
  done
done | tee gpurun_out/acc_ab.txt
timeout 600 python profiles/fin_phases.py c2 c4 > gpurun_out/fin_phases_acc.txt 2>&1; cat gpurun_out/fin_phases_acc.txt
