timeout 1500 python profiles/shard_probe.py c2 cold 2>&1 | grep "N=" | tee gpurun_out/r05_shard_c2_cold.txt
timeout 1500 python profiles/shard_probe.py c4 steady 2>&1 | grep "N=" | tee gpurun_out/r05_shard_c4_steady.txt
