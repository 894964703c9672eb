mkdir -p gpurun_out/r06
timeout 900 python profiles/ring_sort_probe.py 2>&1 | tee gpurun_out/r06/ring_sort_probe.txt
