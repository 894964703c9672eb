timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_loop_kernel.py -q -x 2>&1 | tail -3
for e in 0 1; do echo "== SAGEICP_EXACT_DIVIDE=$e"; SAGEICP_EXACT_DIVIDE=$e SWEEP_LW=2 SWEEP_NW=0 SWEEP_CONTIG=0 timeout 900 python profiles/loop_sweep.py c2 cold 1 8 2>&1 | grep -E "one launch|default"; done
