# two ranks sharing GPU 0 at full scale (60 k queries each on 128 CUs): does the one-launch loop hold with fewer workgroups?
for W in 832 768 704; do
SAGEICP_LOOP_MAX_WGS=$W SAGEICP_LOOP_DEBUG=1 SAGEICP_BENCH_DEVICE=0 SAGEICP_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > /tmp/o.json 2> /tmp/o.err
python - <<PY
import json
d=json.loads([l for l in open('/tmp/o.json') if l.startswith('{')][-1])
print("MAX_WGS=$W", d['value'], d['ms_per_step'], [ (r['loop_form'], r['loop_timeouts']) for r in d['config']['per_rank']])
PY
grep -m2 "one-launch\|loop plan\|plan" /tmp/o.err | cut -c1-200
done
