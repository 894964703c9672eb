# size-classed voxel storage: suite, memory of the c5 / c4 maps with and without classes, bench lines
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gputests_run18.txt 2>&1; grep -n "passed\|failed" gpurun_out/gputests_run18.txt; grep -n "Error\|assert" gpurun_out/gputests_run18.txt | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 600 python profiles/map_memory.py c5 c4 2>&1 | grep "device memory"
SAGEICP_SIZE_CLASSES=0 timeout 600 python profiles/map_memory.py c5 c4 2>&1 | grep "device memory"
for a in "c2 cold" "c2 steady" "c4 steady" "c1 cold" "c5 dense"; do set -- $a
  timeout 600 python bench.py --workload $1 --params $2 --no-cpu-baseline --steps 10 2>/dev/null | grep '^{' > gpurun_out/bench18_$1_$2.json; python -c "
import json; d=json.load(open('gpurun_out/bench18_$1_$2.json')); print('$1 $2:', d['value'], d['ms_per_step'], d['config']['iterations_per_frame'], d['roofline']['avg_launch_us'])"
done
SAGEICP_SIZE_CLASSES=0 timeout 600 python bench.py --workload c2 --params cold --no-cpu-baseline --steps 10 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c2 cold, one class:', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
timeout 300 python profiles/stream_probe.py > gpurun_out/stream_run18.txt 2>&1; grep "per frame" gpurun_out/stream_run18.txt
STREAM_LOCALMAP=1 timeout 300 python profiles/stream_probe.py > gpurun_out/stream_localmap_run18.txt 2>&1; grep "per frame\|LocalMap() per" gpurun_out/stream_localmap_run18.txt
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stream_kt18 -o kt -- python profiles/stream_probe.py > gpurun_out/stream_kt18.txt 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/stream_kt18/**/kt_kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
for r in rows[:28]:
    print('%-70s %6s calls  avg %8.2f us  total %8.2f ms' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
