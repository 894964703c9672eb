#!/bin/bash
# LocalMap() after every streamed frame: how many host threads should populate the destination?
mkdir -p gpurun_out
( nproc; for t in 0 2 4 8 16; do echo "== SAGEICP_TOUCH_THREADS=$t"; SAGEICP_TOUCH_THREADS=$t STREAM_LOCALMAP=1 timeout 600 python profiles/stream_probe.py 2>&1 | grep -E "LocalMap\(\) per frame|per frame ms"; done ) > gpurun_out/r04_touch_threads.txt 2>&1
cat gpurun_out/r04_touch_threads.txt
