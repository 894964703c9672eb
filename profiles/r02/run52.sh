# one scan instead of two for seeded queries, low-pressure need mask
python profiles/knob_probe.py "" ""
for w in "c1" "c5" "c4"; do
  python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', d['value'], 'frames/s', d['ms_per_step'], 'ms', d['config']['iterations_per_frame'], 'it', d['roofline']['avg_launch_us'], 'us/k_icp', d['roofline']['pairs_evaluated_frac'])"
done
python profiles/stream_probe.py 2>&1 | grep -E "per frame"
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -4
