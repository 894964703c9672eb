#!/bin/bash
# flat-order scan in k_loop (8+ lanes per query): exactness, then A/B against the build without it
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_loop_kernel.py -x -q -m gpu 2>&1 | tail -5
NF=sage-icp_amd/_probe/libsageicp_noflat.so
( for rep in 1 2; do
  echo "== flat order (product build), repetition $rep"; timeout 600 python profiles/loop_probe.py quick 2>&1 | grep -E "queries|one launch, LW=(3|4), >=4|launch per"
  echo "== per-voxel restart (-DSAGE_LOOP_FLAT=0), repetition $rep"; LOOP_LIB=$NF timeout 600 python profiles/loop_probe.py quick 2>&1 | grep -E "queries|one launch, LW=(3|4), >=4|launch per"
  done
  echo "== stream, flat"; STREAM_PREFETCH=1 timeout 600 python profiles/stream_probe.py 2>&1 | grep -E "per frame ms|frames "
  echo "== stream, per-voxel restart"; LOOP_LIB=$NF STREAM_PREFETCH=1 timeout 600 python profiles/stream_probe.py 2>&1 | grep -E "per frame ms|frames "
  echo "== timeline 15k, flat"; timeout 300 python profiles/loop_times.py 8 cold | grep -E "^mean|->|scan|body"
) > gpurun_out/r04_flat_ab.txt 2>&1
cat gpurun_out/r04_flat_ab.txt
