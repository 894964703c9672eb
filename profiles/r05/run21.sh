mkdir -p gpurun_out
timeout 900 python profiles/shard_probe.py c4 steady > gpurun_out/r05_shard_c4_steady.txt 2>&1; cat gpurun_out/r05_shard_c4_steady.txt
timeout 600 python profiles/shard_probe.py c2 cold > gpurun_out/r05_shard_c2_cold.txt 2>&1; cat gpurun_out/r05_shard_c2_cold.txt
