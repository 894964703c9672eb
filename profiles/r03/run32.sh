for i in 1 2 3; do STREAM_PREFETCH=1 timeout 300 python profiles/stream_probe.py 2>&1 | grep "per frame"; done
timeout 300 python profiles/stream_probe.py 2>&1 | grep "per frame"
nproc; cat /proc/loadavg
