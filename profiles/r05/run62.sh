# soak of the one-launch loop on the final tree: no launch may give up, every pose the same to the bit
timeout 900 python profiles/loop_soak.py 90 2>&1 | grep "registrations" | tee gpurun_out/r05_loop_soak.txt
