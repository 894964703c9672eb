# state after the accumulators: suite, stream with LocalMap() every frame (pinned pieces), c4 cold, shard probes
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gputests_run07.txt 2>&1; grep -n "passed\|failed" gpurun_out/gputests_run07.txt; grep -n "Error\|assert" gpurun_out/gputests_run07.txt | head -20
STREAM_LOCALMAP=1 timeout 300 python profiles/stream_probe.py > gpurun_out/stream_localmap_run07.txt 2>&1; cat gpurun_out/stream_localmap_run07.txt
timeout 600 python bench.py --workload c4 --params cold --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c4 cold', d['value'], d['ms_per_step'], d['config']['iterations_per_frame'], d['roofline']['avg_launch_us'])" | tee gpurun_out/c4_cold_run07.txt
timeout 600 python profiles/shard_probe.py c2 cold > gpurun_out/shard_c2_run07.txt 2>&1; cat gpurun_out/shard_c2_run07.txt
timeout 900 python profiles/shard_probe.py c4 steady > gpurun_out/shard_c4_run07.txt 2>&1; cat gpurun_out/shard_c4_run07.txt
