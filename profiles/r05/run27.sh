timeout 1500 python -m pytest tests/test_loop_kernel.py -x -q 2>&1 | tail -4
SWEEP_LW=2,3 SWEEP_NW=0 SWEEP_CONTIG=0 timeout 900 python profiles/loop_sweep.py c2 cold 1 8 2>&1 | grep -E "one launch|library|default"
SWEEP_LW=3,4 SWEEP_NW=0 SWEEP_CONTIG=0 timeout 900 python profiles/loop_sweep.py c1 cold 1 20 2>&1 | grep -E "one launch|library|default"
