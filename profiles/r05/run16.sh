mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_loop_kernel.py -x -q 2>&1 | tail -8 > gpurun_out/r05_run16_looptests.txt
cat gpurun_out/r05_run16_looptests.txt
for mix in 1 0; do
SAGEICP_LOOP_MIX=$mix SWEEP_LW=2,3 SWEEP_NW=0 SAGEICP_LOOP_DEBUG=1 timeout 900 python profiles/loop_sweep.py c2 cold 1 5 > gpurun_out/r05_run16_sweep_c2_mix$mix.txt 2>&1
echo "mix $mix"; grep -E "library default|one launch" gpurun_out/r05_run16_sweep_c2_mix$mix.txt; grep "^sageicp" gpurun_out/r05_run16_sweep_c2_mix$mix.txt | sort | uniq -c
done
