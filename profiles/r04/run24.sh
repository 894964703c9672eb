#!/bin/bash
# flat-order scan: where does it pay?  product build (k_loop from 8 lanes per query), variant A (k_loop from 2, k_icp
# from 8), variant B (both from 2)
mkdir -p gpurun_out
P=sage-icp_amd/_probe
( for lib in "" $P/libsageicp_flatA.so $P/libsageicp_flatB.so; do
    export SAGEICP_VARIANT_LIB=$lib LOOP_LIB=$lib
    echo "=================== ${lib:-product build}"
    timeout 900 python profiles/loop_probe.py quick 2>&1 | grep -E "queries|one launch|launch per"
    for wl in "c2 cold" "c4 cold" "c5 dense" "c2 steady"; do set -- $wl
      timeout 600 python bench.py --workload $1 --params $2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$wl', d['value'], 'frames/s', d['ms_per_step'], 'ms', d['config'].get('lanes_per_query'), d['roofline'].get('loop_form'))"
    done
    echo "-- stream (prefetch)"; STREAM_PREFETCH=1 timeout 600 python profiles/stream_probe.py 2>&1 | grep -E "per frame ms"
  done ) > gpurun_out/r04_flat_where.txt 2>&1
cat gpurun_out/r04_flat_where.txt
