# r03 session start: the GPU suite as inherited from round 2 + k_fin phase stamps (instrumented build) on c2 and c4
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 600 python profiles/fin_phases.py c2 c4 > gpurun_out/fin_phases_r02code.txt 2>&1; cat gpurun_out/fin_phases_r02code.txt
