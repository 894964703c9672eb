# lanes per query on the small frames, after this round's changes: shards of c2 / c4 and the stream
for lw in 2 3 4; do echo "SAGEICP_LW=$lw c2 shards"; SAGEICP_LW=$lw timeout 600 python profiles/shard_probe.py c2 cold 2>&1 | tail -3; done
for lw in 1 2 3; do echo "SAGEICP_LW=$lw c4 shards"; SAGEICP_LW=$lw timeout 900 python profiles/shard_probe.py c4 steady 2>&1 | tail -3; done
for lw in 2 3 4; do echo "SAGEICP_LW=$lw stream"; SAGEICP_LW=$lw timeout 300 python profiles/stream_probe.py 2>&1 | grep "per frame"; done
echo "plain stream / localmap / prefetch"
timeout 300 python profiles/stream_probe.py > gpurun_out/stream_final3.txt 2>&1; grep "per frame" gpurun_out/stream_final3.txt
STREAM_LOCALMAP=1 timeout 300 python profiles/stream_probe.py > gpurun_out/stream_localmap_final3.txt 2>&1; grep "per frame\|LocalMap() per" gpurun_out/stream_localmap_final3.txt
STREAM_PREFETCH=1 timeout 300 python profiles/stream_probe.py > gpurun_out/stream_prefetch_final3.txt 2>&1; grep "per frame" gpurun_out/stream_prefetch_final3.txt
