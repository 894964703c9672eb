# compute side of the c4 strong-scaling curve (one rank's share against the full 10M-pt map)
python profiles/shard_probe.py c4 2>&1 | grep "^N=" > gpurun_out/shard_c4.txt; cat gpurun_out/shard_c4.txt
