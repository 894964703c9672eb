cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
(timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02b/pytest.txt 2>&1); tail -15 gpurun_out/r02b/pytest.txt
(timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r02b/bench.json 2> gpurun_out/r02b/bench.err); cat gpurun_out/r02b/bench.json; tail -3 gpurun_out/r02b/bench.err
(timeout 300 python profiles/shard_probe.py > gpurun_out/r02b/shard.txt 2>&1); cat gpurun_out/r02b/shard.txt
