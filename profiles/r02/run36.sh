# k_fin: done flag fetched with the partials; full-record scan with its own offset unit
python profiles/knob_probe.py "" ""
PHASE_LIB=variants/tGN.so PHASE_KIND=GN python profiles/phase_probe.py 1 8
for w in "c1" "c5"; do
  python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', d['value'], 'frames/s', d['ms_per_step'], 'ms', d['config']['iterations_per_frame'], 'it', d['roofline']['avg_launch_us'], 'us/k_icp')"
done
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5
