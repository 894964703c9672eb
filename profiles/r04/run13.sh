# whole GPU suite after the k_skip removal + overflow fallback; phases of a wave's iteration in k_loop
mkdir -p gpurun_out
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/r04_gputests_run13.txt
cat gpurun_out/r04_gputests_run13.txt
(timeout 300 python profiles/loop_times.py 8 cold | tail -16; timeout 300 python profiles/loop_times.py 8 steady | tail -13) > gpurun_out/r04_loop_phases.txt 2>&1
cat gpurun_out/r04_loop_phases.txt
