# soak again after the host's idle-stream exit of the chained loop: it must never fire on a healthy run
mkdir -p gpurun_out/r06
timeout 1200 python profiles/loop_soak.py 60 2>&1 | tee gpurun_out/r06/loop_soak2.txt
