#!/bin/bash
mkdir -p gpurun_out
( for f in 0 1; do for cfg in "c5 dense" "c5 dense_nosem" "c1 cold"; do set -- $cfg
  SAGEICP_FILTER=$f timeout 600 python bench.py --workload $1 --params $2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1 $2 compact scan=$f:', d['value'], 'frames/s', d['ms_per_step'], 'ms', d['roofline'].get('avg_launch_us'), 'us/launch')"
done; done ) > gpurun_out/r04_filter_sparse.txt 2>&1
cat gpurun_out/r04_filter_sparse.txt
