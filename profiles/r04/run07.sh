# k_loop on the stream (24k-point sources), c1 / c5 / c2 through bench.py, c2 forced into one launch
mkdir -p gpurun_out
for m in 0 1; do echo "== SAGEICP_LOOP=$m"; SAGEICP_LOOP=$m timeout 300 python profiles/stream_probe.py 2>&1 | grep -v "^$" | tail -4; done > gpurun_out/r04_stream_loop.txt
cat gpurun_out/r04_stream_loop.txt
for m in 0 1; do for wl in "c1 cold" "c5 dense" "c2 cold" "c2 steady"; do set -- $wl; echo "== SAGEICP_LOOP=$m $1 $2"; SAGEICP_LOOP=$m timeout 600 python bench.py --workload $1 --params $2 --no-cpu-baseline --steps 10 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'frames/s', d['ms_per_step'], 'ms', d['config'].get('iterations_per_frame'), 'iterations')"; done; done > gpurun_out/r04_bench_loop.txt 2>&1
cat gpurun_out/r04_bench_loop.txt
echo "== c2 cold forced into one launch (2 lanes per query)"; SAGEICP_LOOP=2 timeout 600 python bench.py --workload c2 --params cold --no-cpu-baseline --steps 10 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'frames/s', d['ms_per_step'], 'ms', d['roofline']['lanes_per_query'] if d.get('roofline') else '')"
