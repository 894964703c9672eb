# round 6, run 04: per-wave timeline of the first units (which SIMD, which unit, when)
mkdir -p gpurun_out/r06
timeout 600 python profiles/loop_tail.py 1 cold c2 2>&1 | tee gpurun_out/r06/loop_tail_c2_waves.txt | tail -22
