# round 6, run 03: where an iteration waits — per-workgroup timeline (instrumented build), c2 cold / c1
mkdir -p gpurun_out/r06
timeout 600 python profiles/loop_tail.py 1 cold c2 2>&1 | tee gpurun_out/r06/loop_tail_c2.txt
timeout 600 python profiles/loop_tail.py 1 cold c1 2>&1 | tee gpurun_out/r06/loop_tail_c1.txt
