# four ranks sharing GPU 0 (64 CUs each): how many workgroups are resident behind a quarter mask?
for W in 352 320 288 256; do
SAGEICP_LOOP_MAX_WGS=$W SAGEICP_LOOP_DEBUG=1 SAGEICP_BENCH_DEVICE=0 SAGEICP_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 4 --steps 10 --warmup 3 --no-cpu-baseline > /tmp/o.json 2> /tmp/o.err
python - <<PY
import json
d=json.loads([l for l in open('/tmp/o.json') if l.startswith('{')][-1])
print("MAX_WGS=$W", d['value'], d['ms_per_step'], [ (r['loop_form'], r['loop_timeouts']) for r in d['config']['per_rank']])
PY
grep -m1 "one-launch loop for" /tmp/o.err | cut -c1-220
done
