# the compact-scan (fp32 filter) choice re-measured: shards of c2 / c4, the stream, c5, c1
for f in 0 1; do echo "SAGEICP_FILTER=$f c2 shards"; SAGEICP_FILTER=$f timeout 600 python profiles/shard_probe.py c2 cold 2>&1 | tail -4; done
for f in 0 1; do echo "SAGEICP_FILTER=$f c4 shards"; SAGEICP_FILTER=$f timeout 900 python profiles/shard_probe.py c4 steady 2>&1 | tail -4; done
for f in 0 1; do echo "SAGEICP_FILTER=$f stream"; SAGEICP_FILTER=$f timeout 300 python profiles/stream_probe.py 2>&1 | grep "per frame"; done
for a in "c5 dense" "c1 cold"; do set -- $a; for f in 0 1; do
  SAGEICP_FILTER=$f timeout 600 python bench.py --workload $1 --params $2 --no-cpu-baseline --steps 8 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 $2 filter=$f:', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
done; done
