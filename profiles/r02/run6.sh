cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02f
SAGEICP_BENCH_DEVICE=0 SAGEICP_NO_P2P=1 MASTER_ADDR=127.0.0.1 timeout 300 python bench.py --gpus 2 --steps 2 --warmup 1 --scale 0.1 --no-cpu-baseline > gpurun_out/r02f/rccl2.out 2> gpurun_out/r02f/rccl2.err; echo rc=$?; cat gpurun_out/r02f/rccl2.out; grep -v "^\[W\|amdgpu.ids" gpurun_out/r02f/rccl2.err | tail -15
timeout 300 python bench.py > gpurun_out/r02f/bench.json 2> gpurun_out/r02f/bench.err; cat gpurun_out/r02f/bench.json; tail -3 gpurun_out/r02f/bench.err
