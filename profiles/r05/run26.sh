SWEEP_LW=2 SWEEP_NW=0 SWEEP_CONTIG=0 timeout 900 python profiles/loop_sweep.py c2 cold 1 8 2>&1 | grep -E "one launch|library|default"
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "index or exact or near or tie or filter or compact" 2>&1 | tail -3
