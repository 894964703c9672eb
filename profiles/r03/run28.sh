for rep in 1 2; do for t in 0 4 8; do echo "THREADS=$t"; SAGEICP_TOUCH_THREADS=$t STREAM_LOCALMAP=1 timeout 300 python profiles/stream_probe.py 2>&1 | grep "LocalMap() per"; done; done
