SWEEP_LW=1,2 SWEEP_NW=0 SWEEP_CONTIG=0 timeout 900 python profiles/loop_sweep.py c2 cold 1 6 2>&1 | grep -E "one launch|default|LW=" | cut -c1-160
