# k_loop with counted accumulator words (no ack wait, no arrival counter) and the LDS-only ticket (both loops)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_loop_kernel.py tests/test_gpu_parity.py -m gpu -x -q -k "not rccl_refuses" 2>&1 | tail -6 > gpurun_out/r04_tests_run14.txt
cat gpurun_out/r04_tests_run14.txt
(timeout 300 python profiles/loop_times.py 8 cold | grep -v "^   [ 0-9]* |" | grep -v "slowest\|per XCD\|correlation\|quintile") > gpurun_out/r04_loop_times4.txt 2>&1
cat gpurun_out/r04_loop_times4.txt
timeout 900 python profiles/loop_probe.py quick > gpurun_out/r04_loop_probe2.txt 2>&1
grep -A3 "queries" gpurun_out/r04_loop_probe2.txt | grep -v "LW=2\|LW=4" 
for m in 0 1; do echo "== SAGEICP_LOOP=$m"; SAGEICP_LOOP=$m timeout 300 python profiles/stream_probe.py 2>&1 | grep "per frame"; done
timeout 600 python bench.py --no-cpu-baseline --steps 10 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c2 cold', d['value'], d['ms_per_step'])"
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -rP -k "rccl_refuses" 2>&1 | grep -i "sageicp\|nccl\|duplicate\|invalid" | head -20
