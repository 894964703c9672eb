# final tree (adaptive poll back-off): GPU suite, profile sets
(timeout 3000 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -3) | tee gpurun_out/r05_gputests.txt
for w in "c2 cold" "c4 steady" "c5 dense" "c1 cold"; do bash profiles/run_profiles.sh r05 $w > /dev/null 2>&1; echo "== $w"; tail -1 gpurun_out/prof_r05_${w% *}-${w#* }/summary.md | cut -c1-100; done
