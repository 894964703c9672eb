# the ICP loop as a hipGraph of N iterations: bench c2 / c1 / stream for N = 0 (plain stream launches), 2, 4, 8
for g in 0 4 2 8 0 4; do echo "SAGEICP_GRAPH_ITERS=$g"; SAGEICP_GRAPH_ITERS=$g python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('  c2 %.1f frames/s  %.3f ms  k_icp %.1f us' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_us']))"
done > gpurun_out/graph_loop.txt 2>&1
for g in 0 4 8; do echo "SAGEICP_GRAPH_ITERS=$g c1"; SAGEICP_GRAPH_ITERS=$g python bench.py --workload c1 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('  c1 %.1f frames/s  %.3f ms' % (d['value'], d['ms_per_step']))"
done >> gpurun_out/graph_loop.txt 2>&1
for g in 0 4; do echo "SAGEICP_GRAPH_ITERS=$g stream"; SAGEICP_GRAPH_ITERS=$g STREAM_PREFETCH=1 python profiles/stream_probe.py 2>&1 | grep "per frame"; done >> gpurun_out/graph_loop.txt 2>&1
cat gpurun_out/graph_loop.txt
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -6
