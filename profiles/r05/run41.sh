timeout 1500 python profiles/solver_cu_probe.py 2>&1 | grep "solver CU" | tee gpurun_out/r05_solver_cu.txt
