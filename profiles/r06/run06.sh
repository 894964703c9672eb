# round 6, run 06: the timeline with the heaviest units at the highest priority and first units dealt by SIMD
mkdir -p gpurun_out/r06
SAGEICP_LOOP_PRIO=4 SAGEICP_LOOP_DEAL=1 timeout 600 python profiles/loop_tail.py 1 cold c2 2>&1 | tee gpurun_out/r06/loop_tail_c2_prio4_deal.txt | grep -v "last ten"
