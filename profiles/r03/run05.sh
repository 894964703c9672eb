# replay with the exact fast growth, prefetch fingerprint / cancel, multi-device set-up gate: suite + stream
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gputests_run05.txt 2>&1; grep -n "passed\|failed" gpurun_out/gputests_run05.txt; grep -n "Error\|assert" gpurun_out/gputests_run05.txt | head -20
timeout 300 python profiles/stream_probe.py > gpurun_out/stream_run05.txt 2>&1; cat gpurun_out/stream_run05.txt
SAGEICP_DEBUG_ORDER=1 timeout 300 python profiles/stream_probe.py 2>&1 | grep "order level" | tail -3
STREAM_PREFETCH=1 timeout 300 python profiles/stream_probe.py 2>&1 | grep "per frame"
