#!/bin/bash
mkdir -p gpurun_out
( for rep in 1 2; do for w in 4 8; do echo "== SAGEICP_LOOP_WAVES=$w (rep $rep)"; SAGEICP_LOOP_WAVES=$w STREAM_PREFETCH=1 timeout 600 python profiles/stream_probe.py 2>&1 | grep -E "per frame ms"; done; done ) > gpurun_out/r04_stream_waves.txt 2>&1
cat gpurun_out/r04_stream_waves.txt
