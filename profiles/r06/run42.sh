# small frames searched as they came (kSortFrameFrom): GPU suite, then the A/B by size against "always sorted"
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/gputests_run42.txt 2>&1; grep -n "passed\|failed" gpurun_out/gputests_run42.txt | tail -3
mkdir -p gpurun_out/r06
timeout 1200 python profiles/knob_ab.py "c1:cold:1:60 c1:steady:1:60 c2:cold:16:60 c2:cold:8:40 c2:cold:4:30" "SAGEICP_SORT_FROM=0" "" "SAGEICP_SORT_FROM=40000" 2>&1 | tee gpurun_out/r06/sort_from_ab.txt
timeout 300 python bench.py --workload c1 --params cold --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | tail -1 | cut -c1-300
