# short scans peeled out of the double-step loop
python profiles/knob_probe.py "" ""
for w in "c1" "c5" "c4"; do
  python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', d['value'], 'frames/s', d['ms_per_step'], 'ms', d['config']['iterations_per_frame'], 'it', d['roofline']['avg_launch_us'], 'us/k_icp')"
done
