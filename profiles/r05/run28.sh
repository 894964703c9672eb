mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/r05_gputests.txt
cat gpurun_out/r05_gputests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
for wl in "c1 cold" "c2 cold" "c2 steady" "c4 steady" "c4 cold" "c5 dense" "c5 dense_nosem"; do set -- $wl; timeout 900 python bench.py --workload $1 --params $2 --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/r05_bench_$1_$2.json; python -c "
import json; d=json.load(open('gpurun_out/r05_bench_$1_$2.json')); print('$1 $2', d['value'], 'fps', d['ms_per_step'], 'ms', d['config']['iterations_per_frame'], 'it', d['roofline']['loop_form'][:12], d['roofline']['lanes_per_query'], 'lanes', d['roofline']['avg_launch_us'], 'us/it')"; done
timeout 600 python profiles/loop_times.py 1 cold c2 > gpurun_out/r05_loop_times_c2.txt 2>&1; sed -n 1,12p gpurun_out/r05_loop_times_c2.txt
