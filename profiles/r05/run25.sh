SWEEP_LW=2 SWEEP_NW=0 SWEEP_CONTIG=0,1,2 timeout 900 python profiles/loop_sweep.py c2 cold 1 8 2>&1 | grep -E "one launch|library"
SWEEP_LW=2 SWEEP_NW=0 SWEEP_CONTIG=0,1,2 timeout 900 python profiles/loop_sweep.py c2 steady 1 8 2>&1 | grep -E "one launch|library"
