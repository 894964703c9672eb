# the compute side of the strong-scaling curve on the round-6 kernels (one rank's share, no exchange)
mkdir -p gpurun_out/r06
timeout 900 python profiles/shard_probe.py c2 cold 2>&1 | tee gpurun_out/r06/shard_c2_cold.txt
timeout 900 python profiles/shard_probe.py c4 steady 2>&1 | tee gpurun_out/r06/shard_c4_steady.txt
