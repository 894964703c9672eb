# host look-ahead depth of the polled loop on small frames (trailing no-op launches vs keeping the GPU fed)
for d in 2 3 4 6; do
  SAGEICP_DEPTH=$d timeout 300 python bench.py --workload c1 --no-cpu-baseline --steps 30 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c1 depth=$d', d['value'], d['ms_per_step'], d['config']['iterations_per_frame'])"
  SAGEICP_DEPTH=$d timeout 300 python profiles/stream_probe.py 2>&1 | grep "per frame"
  SAGEICP_DEPTH=$d timeout 300 python bench.py --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c2 depth=$d', d['value'], d['ms_per_step'])"
done | tee gpurun_out/depth_probe.txt
