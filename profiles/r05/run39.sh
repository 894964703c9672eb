# profile sets of this tree: the headline, the big frame, the dense map, the small frame; then the parity suites
for w in "c2 cold" "c4 steady" "c5 dense" "c1 cold"; do bash profiles/run_profiles.sh r05 $w > /dev/null 2>&1; echo "== $w"; tail -4 gpurun_out/prof_r05_${w% *}-${w#* }/summary.md | cut -c1-150; done
(timeout 3000 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -3) > gpurun_out/r05_gputests.txt
cat gpurun_out/r05_gputests.txt
