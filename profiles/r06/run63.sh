# c4 (500k vs 10M) on two ranks sharing GPU 0 (128 CUs each, direct exchange between the processes): launches chained beside
# each rank's solving wave against k_fin between them
mkdir -p gpurun_out/r06
for C in 1 0; do
SAGEICP_CHAIN_COMM=1 SAGEICP_CHAIN=$C SAGEICP_BENCH_DEVICE=0 SAGEICP_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --workload c4 --params steady --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r06/bench_c4_shared_n2_chain$C.json 2> gpurun_out/r06/bench_c4_shared_n2_chain$C.err
python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r06/bench_c4_shared_n2_chain$C.json') if l.startswith('{')][-1])
print("CHAIN=$C", d['value'], d['ms_per_step'], d['config']['exchange'], d['config']['iterations_per_frame'], [(r['calls_chained'], r['calls_per_iteration'], r['loop_timeouts']) for r in d['config']['per_rank']], d['config'].get('pose_error_vs_planted'))
PY
done
