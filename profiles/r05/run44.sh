# final tree: whole GPU suite, smoke, four profile sets, the bench lines (after collect.py these carry roofline.frac only
# if run again: run45)
(timeout 3000 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -3) > gpurun_out/r05_gputests.txt
cat gpurun_out/r05_gputests.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
for w in "c2 cold" "c4 steady" "c5 dense" "c1 cold"; do bash profiles/run_profiles.sh r05 $w > /dev/null 2>&1; echo "== $w"; tail -2 gpurun_out/prof_r05_${w% *}-${w#* }/summary.md | cut -c1-120; done
