# where a small frame's iteration goes (timing build): c1 whole, c2 / 8 (a shard of eight)
(timeout 600 python profiles/loop_times.py 1 cold c1 2>&1 | grep -v "^$" | head -60) > gpurun_out/r05_loop_times_c1.txt
(timeout 600 python profiles/loop_times.py 8 cold c2 2>&1 | grep -v "^$" | head -60) > gpurun_out/r05_loop_times_c2_div8.txt
grep -E "mean:|->|pose, query|row rebuild|seed,|bounds|scan|argmin|answer|pair terms|body|closing|waiting|queries," gpurun_out/r05_loop_times_c1.txt gpurun_out/r05_loop_times_c2_div8.txt
