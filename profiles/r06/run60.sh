# soak of the round-6 kernels: every registration in one launch (or chained), every pose the same bits
mkdir -p gpurun_out/r06
timeout 1500 python profiles/loop_soak.py 90 2>&1 | tee gpurun_out/r06/loop_soak.txt
