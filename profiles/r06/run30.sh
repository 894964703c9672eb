# round 6, run 30: what the streamed frame's preprocessing is made of
mkdir -p gpurun_out/r06
SAGEICP_DEBUG_ORDER=1 timeout 600 python profiles/stream_probe.py 2>&1 | grep -E "order level|per frame|frames" | tail -12 | tee gpurun_out/r06/stream_order_debug.txt
