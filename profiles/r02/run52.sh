# hipGraph launch-floor probe + the GPU suite on the re-created container's build
hipcc --offload-arch=gfx950 -O2 profiles/graph_floor.hip -o /tmp/gf && timeout 120 /tmp/gf > gpurun_out/graph_floor.txt 2>&1
cat gpurun_out/graph_floor.txt
( time timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 ) > gpurun_out/gputests.txt 2>&1
cat gpurun_out/gputests.txt
