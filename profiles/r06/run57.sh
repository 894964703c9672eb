# phase split of a pass of k_loop on the final tree (instrumented build): c2 full frame cold + steady, c1
mkdir -p gpurun_out/r06
(timeout 600 python profiles/loop_phases.py 1 cold c2; timeout 600 python profiles/loop_phases.py 1 steady c2; timeout 600 python profiles/loop_phases.py 1 cold c1) 2>&1 | grep -v "^$" | tee gpurun_out/r06/loop_phases.txt
