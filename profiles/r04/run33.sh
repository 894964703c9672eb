#!/bin/bash
# k_icp with three / four pairs of candidates in flight per lane and the registers for it (5 / 4 waves per SIMD):
# fewer, fatter waves on the big frames?
mkdir -p gpurun_out
P=sage-icp_amd/_probe
( for lib in "" $P/libsageicp_d3o5.so $P/libsageicp_d4o4.so; do
    export SAGEICP_VARIANT_LIB=$lib
    echo "=================== ${lib:-product build (2 pairs in flight, 7 waves per SIMD)}"
    for cfg in "c2 cold 1" "c2 cold 2" "c2 steady 1" "c2 steady 2" "c4 cold 0" "c4 cold 1" "c4 cold 2" "c5 dense 0" "c5 dense 1"; do set -- $cfg
      SAGEICP_LW=$3 timeout 600 python bench.py --workload $1 --params $2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1 $2 lw=$3:', d['value'], 'frames/s', d['ms_per_step'], 'ms', d['roofline'].get('avg_launch_us'), 'us/launch')"
    done
  done ) > gpurun_out/r04_icp_depth.txt 2>&1
cat gpurun_out/r04_icp_depth.txt
