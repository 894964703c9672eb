# compact candidates (immediate fetch), 6 waves per SIMD: every workload, with the filter on and off
for f in 0 1; do
  export SAGEICP_NO_FILTER=$f; echo "== SAGEICP_NO_FILTER=$f"
  python profiles/knob_probe.py ""
  for w in "c1" "c4" "c5" "c5 --params dense_nosem" "c2 --params steady"; do
    python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', d['value'], 'frames/s', d['ms_per_step'], 'ms', d['config']['iterations_per_frame'], 'it', d['roofline']['avg_launch_us'], 'us/k_icp')"
  done
  python profiles/stream_probe.py 2>&1 | grep -E "per frame|frames"
done
unset SAGEICP_NO_FILTER
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5
