# stream figures of the round: plain, with the next frame announced, with LocalMap() every frame
mkdir -p gpurun_out
(echo "== plain"; timeout 300 python profiles/stream_probe.py 2>&1 | grep -v "^$" | tail -4
 echo "== next frame announced (sageicp_pipeline_prefetch)"; STREAM_PREFETCH=1 timeout 300 python profiles/stream_probe.py 2>&1 | grep -v "^$" | tail -4
 echo "== LocalMap() after every frame"; STREAM_LOCALMAP=1 timeout 300 python profiles/stream_probe.py 2>&1 | grep -v "^$" | tail -5) > gpurun_out/r04_stream_final.txt 2>&1
cat gpurun_out/r04_stream_final.txt
