( echo "== plain"; timeout 300 python profiles/stream_probe.py 2>&1 | tail -4
echo "== next frame announced"; STREAM_PREFETCH=1 timeout 300 python profiles/stream_probe.py 2>&1 | tail -4
echo "== LocalMap() after every frame"; STREAM_LOCALMAP=1 timeout 300 python profiles/stream_probe.py 2>&1 | tail -5 ) > gpurun_out/r05_stream.txt
cat gpurun_out/r05_stream.txt
