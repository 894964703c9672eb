#!/bin/bash
# lanes per query x order of the scan on the big frames (product build; 8 lanes: always flat)
mkdir -p gpurun_out
( for cfg in "c2 cold 2 0" "c2 cold 2 1" "c2 cold 3 1" "c2 steady 2 0" "c2 steady 3 1" "c4 cold 1 0" "c4 cold 1 1" "c4 cold 2 0" "c4 cold 2 1" "c5 dense 1 0" "c5 dense 1 1" "c5 dense 2 0" "c5 dense 2 1" "c5 dense 0 0"; do set -- $cfg
    SAGEICP_LW=$3 SAGEICP_FLAT=$4 timeout 600 python bench.py --workload $1 --params $2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1 $2 lw=$3 flat=$4:', d['value'], 'frames/s', d['ms_per_step'], 'ms', d['roofline'].get('avg_launch_us'), 'us/launch')"
  done ) > gpurun_out/r04_flat_big.txt 2>&1
cat gpurun_out/r04_flat_big.txt
