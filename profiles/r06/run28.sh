# round 6, run 28: break-points on the ring-geometry scene family
mkdir -p gpurun_out/r06
timeout 1500 python profiles/ring_probe.py 2>&1 | tee gpurun_out/r06/ring_probe.txt
