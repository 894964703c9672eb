# per-query record of the previous iteration shrunk to 8 bytes
python profiles/knob_probe.py "" ""
for w in "c1" "c5" "c4"; do
  python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', d['value'], 'frames/s', d['ms_per_step'], 'ms', d['roofline']['avg_launch_us'], 'us/k_icp')"
done
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
