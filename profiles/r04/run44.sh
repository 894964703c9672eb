#!/bin/bash
# k_loop: the published pose in 8 / 32 copies (256 B apart), a workgroup polls copy blockIdx % copies — 480 pollers on one line?
mkdir -p gpurun_out
P=sage-icp_amd/_probe
( SAGEICP_VARIANT_LIB=$P/libsageicp_pose8.so timeout 600 python -m pytest tests/test_loop_kernel.py -x -q -m gpu 2>&1 | tail -2
  for rep in 1 2; do for lib in "" $P/libsageicp_pose8.so $P/libsageicp_pose32.so; do
    echo "== ${lib:-product}, repetition $rep"
    LOOP_LIB=$lib timeout 600 python profiles/loop_probe.py quick 2>&1 | grep -E "queries|one launch, LW=(3|4), >=4" | grep -v "launch per iteration, 16"
    LOOP_LIB=$lib STREAM_PREFETCH=1 timeout 600 python profiles/stream_probe.py 2>&1 | grep -E "per frame ms"
  done; done ) > gpurun_out/r04_pose_copies.txt 2>&1
cat gpurun_out/r04_pose_copies.txt
