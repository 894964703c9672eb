# the driver's N>1 command on a 1-GPU box: two ranks sharing GPU 0 (CU-masked streams), direct exchange between processes; then 4 ranks
mkdir -p gpurun_out/r06
for N in 2 4; do
SAGEICP_BENCH_DEVICE=0 SAGEICP_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus $N --steps 10 --warmup 3 > gpurun_out/r06/bench_shared_gpu_n$N.json 2> gpurun_out/r06/bench_shared_gpu_n$N.err
echo "N=$N rc=$?"; tail -c 1800 gpurun_out/r06/bench_shared_gpu_n$N.json | cut -c1-1800; tail -3 gpurun_out/r06/bench_shared_gpu_n$N.err
done
