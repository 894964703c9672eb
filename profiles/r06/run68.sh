# round trip of a word between two waves: same XCD or not, agent scope or through the XCD's own L2 (sc0 only)
mkdir -p gpurun_out/r06
timeout 120 ./profiles/micro/xcd_ping 2>&1 | tee gpurun_out/r06/xcd_ping.txt
