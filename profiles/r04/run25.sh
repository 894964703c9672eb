#!/bin/bash
# flat order as built into the product (8/16 lanes: always; 2/4 lanes: sparse voxels; k_loop takes more lanes): the suite,
# the bench lines, the stream, the shards
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r04_flat_suite.txt; cat gpurun_out/r04_flat_suite.txt
( for wl in "c1 cold" "c2 cold" "c2 steady" "c4 cold" "c4 steady" "c5 dense" "c5 dense_nosem"; do set -- $wl
    timeout 600 python bench.py --workload $1 --params $2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$wl', d['value'], 'frames/s', d['ms_per_step'], 'ms', d['config'].get('lanes_per_query'), d['roofline'].get('loop_form'))"
  done
  echo "-- stream (prefetch)"; STREAM_PREFETCH=1 timeout 600 python profiles/stream_probe.py 2>&1 | grep -E "per frame ms|frames "
  echo "-- stream"; timeout 600 python profiles/stream_probe.py 2>&1 | grep -E "per frame ms"
  timeout 900 python profiles/loop_probe.py quick 2>&1 | grep -E "queries|launch per"
  echo "-- defaults (no knobs)"; timeout 600 python profiles/round_probe.py 2>&1 | tail -15
) > gpurun_out/r04_flat_product.txt 2>&1
cat gpurun_out/r04_flat_product.txt
