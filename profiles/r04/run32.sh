#!/bin/bash
# flat scan: the next voxel's row word read one crossing ahead (-DSAGE_FLAT_AHEAD=1) against the product build
mkdir -p gpurun_out
A=sage-icp_amd/_probe/libsageicp_ahead.so
( SAGEICP_VARIANT_LIB=$A timeout 600 python -m pytest tests/test_loop_kernel.py -x -q -m gpu 2>&1 | tail -2
  for rep in 1 2; do for lib in "" $A; do
    echo "== ${lib:-product}, repetition $rep"
    LOOP_LIB=$lib timeout 600 python profiles/loop_probe.py quick 2>&1 | grep -E "queries|one launch, LW=(3|4), >=4|launch per"
    LOOP_LIB=$lib STREAM_PREFETCH=1 timeout 600 python profiles/stream_probe.py 2>&1 | grep -E "per frame ms"
  done; done ) > gpurun_out/r04_flat_ahead.txt 2>&1
cat gpurun_out/r04_flat_ahead.txt
