# two (and four) ranks sharing GPU 0 at full scale after the residency rule for CU-masked streams
mkdir -p gpurun_out/r06
for N in 2 4; do
SAGEICP_BENCH_DEVICE=0 SAGEICP_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus $N --steps 10 --warmup 3 > gpurun_out/r06/bench_shared_gpu_n$N.json 2> gpurun_out/r06/bench_shared_gpu_n$N.err
python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r06/bench_shared_gpu_n$N.json') if l.startswith('{')][-1])
print("N=$N", d['value'], d['ms_per_step'], d['config']['exchange'], [ (r['loop_form'], r['loop_timeouts']) for r in d['config']['per_rank']], d['config'].get('pose_error_vs_planted'))
PY
done
