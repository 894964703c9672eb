# round 6, run 22: the solving wave's time, split
mkdir -p gpurun_out/r06
timeout 600 python profiles/solve_split.py c1 2>&1 | tee gpurun_out/r06/solve_split.txt
