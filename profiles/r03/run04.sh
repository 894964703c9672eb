# k_red (two-stage reduction for big frames), replay pool + arrival-order source level: suite, c4 / c2 lines, stream
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gputests_run04.txt 2>&1; tail -5 gpurun_out/gputests_run04.txt
timeout 600 python bench.py --workload c4 --no-cpu-baseline --steps 10 > gpurun_out/bench_c4_run04.json 2> gpurun_out/bench_c4_run04.err; python -c "
import json; d=json.load(open('gpurun_out/bench_c4_run04.json')); print('c4', d['value'], d['ms_per_step'], d['config']['iterations_per_frame'], d['roofline']['avg_launch_us'])"
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench_c2_run04.json 2> gpurun_out/bench_c2_run04.err; python -c "
import json; d=json.load(open('gpurun_out/bench_c2_run04.json')); print('c2', d['value'], d['ms_per_step'], d['config']['iterations_per_frame'], d['roofline']['avg_launch_us'])"
timeout 300 python profiles/stream_probe.py > gpurun_out/stream_run04.txt 2>&1; cat gpurun_out/stream_run04.txt
SAGEICP_DEBUG_ORDER=1 timeout 300 python profiles/stream_probe.py 2>&1 | grep "order level" | tail -4
