#!/bin/bash
# k_loop: a workgroup's waves taken from all over the sorted frame (-DSAGE_LOOP_TRANSPOSE=1) instead of consecutive ones:
# every CU gets the frame's average load — against the locality of neighbouring waves
mkdir -p gpurun_out
T=sage-icp_amd/_probe/libsageicp_transpose.so
( SAGEICP_VARIANT_LIB=$T timeout 600 python -m pytest tests/test_loop_kernel.py -x -q -m gpu 2>&1 | tail -2
  for rep in 1 2; do for lib in "" $T; do
    echo "== ${lib:-product}, repetition $rep"
    LOOP_LIB=$lib timeout 600 python profiles/loop_probe.py quick 2>&1 | grep -E "queries|one launch, LW=(3|4), >=4"
    LOOP_LIB=$lib STREAM_PREFETCH=1 timeout 600 python profiles/stream_probe.py 2>&1 | grep -E "per frame ms"
  done; done ) > gpurun_out/r04_loop_transpose.txt 2>&1
cat gpurun_out/r04_loop_transpose.txt
