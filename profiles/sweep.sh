for gm in 32 16 8 4 2 1; do
echo "group_max=$gm"; SAGEICP_GROUP_MAX=$gm timeout 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['avg_launch_us'], r['frac'])"
done
