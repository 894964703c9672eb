# lanes per query revisited for the small frames (c1: 10k points, streamed frames: 24k points)
for lw in 2 3 4; do
  SAGEICP_LW=$lw python bench.py --workload c1 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c1 LW=$lw', d['value'], 'frames/s', d['ms_per_step'], 'ms', d['roofline']['avg_launch_us'], 'us/k_icp')"
  SAGEICP_LW=$lw python profiles/stream_probe.py 2>&1 | grep -E "per frame" | sed "s/^/stream LW=$lw /"
done
