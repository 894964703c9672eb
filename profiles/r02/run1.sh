set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
(timeout 600 python profiles/phase_probe.py 1 4 8 > gpurun_out/r02a/phase.txt 2>&1)
(timeout 300 python profiles/shard_probe.py > gpurun_out/r02a/shard.txt 2>&1)
(timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err)
cat gpurun_out/r02a/phase.txt gpurun_out/r02a/shard.txt gpurun_out/r02a/bench.json
