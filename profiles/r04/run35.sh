#!/bin/bash
# k_icp64 (64 queries per wave: a lane per query for the per-query work, four lanes per query for the scan): parity, then speed
mkdir -p gpurun_out
export SAGEICP_VARIANT_LIB=sage-icp_amd/_probe/libsageicp_fat.so
( SAGEICP_FAT=1 SAGEICP_LOOP=0 timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "register_frame or full_size or streaming or property" 2>&1 | tail -8
  for cfg in "c2 cold" "c2 steady" "c4 cold" "c4 steady" "c5 dense"; do set -- $cfg
    for fat in 0 1; do
      SAGEICP_FAT=$fat timeout 600 python bench.py --workload $1 --params $2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1 $2 fat=$fat:', d['value'], 'frames/s', d['ms_per_step'], 'ms', d['roofline'].get('avg_launch_us'), 'us/launch', 'parity', d.get('parity'))"
    done
  done ) > gpurun_out/r04_fat.txt 2>&1
cat gpurun_out/r04_fat.txt
