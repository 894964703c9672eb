timeout 1500 python -m pytest tests/test_loop_kernel.py -x -q 2>&1 | tail -4
for c in 1 0; do echo "== SAGEICP_LOOP_CENSUS=$c"; SAGEICP_LOOP_CENSUS=$c SWEEP_LW=2 SWEEP_NW=0 SWEEP_CONTIG=0 timeout 900 python profiles/loop_sweep.py c2 cold 1 8 2>&1 | grep -E "one launch|library"; done
SAGEICP_LOOP_CENSUS=1 SWEEP_LW=2 SWEEP_NW=0 SWEEP_CONTIG=0 timeout 900 python profiles/loop_sweep.py c2 steady 1 8 2>&1 | grep -E "one launch|library"
SWEEP_LW=2 SWEEP_NW=0 SWEEP_CONTIG=0 timeout 900 python profiles/loop_sweep.py c2 cold 2 8 2>&1 | grep -E "one launch|library"
