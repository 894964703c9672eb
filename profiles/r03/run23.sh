# round-3 evidence on the code with size-classed voxel storage: suite, smoke, rocprofv3 sets for c2-cold and c4-steady, the driver bench command, other workloads, stream, k_fin phases, shard probes
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gputests_final2.txt 2>&1; grep -n "passed\|failed" gpurun_out/gputests_final2.txt; grep -n "Error\|assert" gpurun_out/gputests_final2.txt | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
bash profiles/run_profiles.sh r03 c2 cold > gpurun_out/prof_c2.log 2>&1; head -14 gpurun_out/prof_r03_c2-cold/summary.md
bash profiles/run_profiles.sh r03 c4 steady > gpurun_out/prof_c4.log 2>&1; head -14 gpurun_out/prof_r03_c4-steady/summary.md
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep '^{' > gpurun_out/bench_driver_cmd.json; python -c "
import json; d=json.load(open('gpurun_out/bench_driver_cmd.json')); print('driver cmd:', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['parity']['ok'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
for a in "c1 cold" "c2 steady" "c4 cold" "c5 dense" "c5 dense_nosem"; do set -- $a
  timeout 600 python bench.py --workload $1 --params $2 --no-cpu-baseline --steps 10 2>/dev/null | grep '^{' > gpurun_out/bench_$1_$2.json; python -c "
import json; d=json.load(open('gpurun_out/bench_$1_$2.json')); print('$1 $2:', d['value'], d['ms_per_step'], d['config']['iterations_per_frame'], d['roofline']['avg_launch_us'])"
done
timeout 300 python profiles/stream_probe.py > gpurun_out/stream_final2.txt 2>&1; grep "per frame" gpurun_out/stream_final2.txt
STREAM_LOCALMAP=1 timeout 300 python profiles/stream_probe.py > gpurun_out/stream_localmap_final2.txt 2>&1; grep "per frame\|LocalMap() per" gpurun_out/stream_localmap_final2.txt
STREAM_PREFETCH=1 timeout 300 python profiles/stream_probe.py > gpurun_out/stream_prefetch_final2.txt 2>&1; grep "per frame" gpurun_out/stream_prefetch_final2.txt
timeout 600 python profiles/fin_phases.py c2 c4 > gpurun_out/fin_phases_final2.txt 2>&1; grep "sum (" gpurun_out/fin_phases_final2.txt
timeout 600 python profiles/shard_probe.py c2 cold > gpurun_out/shard_c2_final2.txt 2>&1; cat gpurun_out/shard_c2_final2.txt
timeout 900 python profiles/shard_probe.py c4 steady > gpurun_out/shard_c4_final2.txt 2>&1; cat gpurun_out/shard_c4_final2.txt
timeout 900 python profiles/map_memory.py c5 c4 lattice 2>&1 | grep "device memory"
