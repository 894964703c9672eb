(timeout 600 python profiles/loop_times.py 1 cold c2 2>&1 | grep -v "^$" | head -70) > gpurun_out/r05_loop_times_c2_final.txt
grep -E "mean:|->|iteration 8|per XCD|quintile|pose, query|row rebuild|seed,|bounds|scan|argmin|answer|pair terms|body|closing|waiting|queries,|workgroups with" gpurun_out/r05_loop_times_c2_final.txt | cut -c1-220
