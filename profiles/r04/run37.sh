#!/bin/bash
# LocalMap(): the destination is a fresh 46-MB allocation every call — glibc serves it by mmap (fresh zero pages, ~11k
# faults).  With malloc told to keep big blocks on its heap (glibc tunables in the environment) the pages of the last
# call's vector are reused.
mkdir -p gpurun_out
( echo "== default malloc"; STREAM_LOCALMAP=1 timeout 600 python profiles/stream_probe.py 2>&1 | grep -E "LocalMap\(\) per frame"
  echo "== MALLOC_MMAP_MAX_=0 MALLOC_TRIM_THRESHOLD_=17179869184"; MALLOC_MMAP_MAX_=0 MALLOC_TRIM_THRESHOLD_=17179869184 STREAM_LOCALMAP=1 timeout 600 python profiles/stream_probe.py 2>&1 | grep -E "LocalMap\(\) per frame|per frame ms"
  echo "== the same, SAGEICP_TOUCH_THREADS=0"; SAGEICP_TOUCH_THREADS=0 MALLOC_MMAP_MAX_=0 MALLOC_TRIM_THRESHOLD_=17179869184 STREAM_LOCALMAP=1 timeout 600 python profiles/stream_probe.py 2>&1 | grep -E "LocalMap\(\) per frame|per frame ms"
  echo "== GLIBC_TUNABLES form"; GLIBC_TUNABLES=glibc.malloc.mmap_max=0:glibc.malloc.trim_threshold=17179869184 STREAM_LOCALMAP=1 timeout 600 python profiles/stream_probe.py 2>&1 | grep -E "LocalMap\(\) per frame"
) > gpurun_out/r04_localmap_malloc.txt 2>&1
cat gpurun_out/r04_localmap_malloc.txt
