# full-record scan with the exit between the two half steps restored
python profiles/knob_probe.py ""
for w in "c1" "c5" "c5 --params dense_nosem"; do
  python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', d['value'], 'frames/s', d['ms_per_step'], 'ms', d['roofline']['avg_launch_us'], 'us/k_icp')"
done
python profiles/stream_probe.py 2>&1 | grep -E "per frame"
