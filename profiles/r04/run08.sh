# non-finite input policy, multi-rank test changes, bench roofline block; stream under k_loop with other lane counts
mkdir -p gpurun_out
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r04_gputests_run08.txt
cat gpurun_out/r04_gputests_run08.txt
for lw in 2 3 4; do for nw in 4 8; do echo "== stream, one launch, SAGEICP_LW=$lw SAGEICP_LOOP_WAVES=$nw"; SAGEICP_LW=$lw SAGEICP_LOOP_WAVES=$nw timeout 300 python profiles/stream_probe.py 2>&1 | grep "per frame"; done; done > gpurun_out/r04_stream_loop_lw.txt 2>&1
cat gpurun_out/r04_stream_loop_lw.txt
timeout 600 python bench.py --no-cpu-baseline --steps 10 2>/dev/null | grep '^{' > gpurun_out/r04_bench_c2_run08.json; python -c "
import json; d=json.load(open('gpurun_out/r04_bench_c2_run08.json')); print(d['value'], d['ms_per_step']); print(json.dumps(d['roofline'], indent=1)[:1500])"
