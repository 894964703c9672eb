mkdir -p gpurun_out
for v in occ8d1 occ7d1; do
LOOP_LIB=sage-icp_amd/_probe/libsageicp_$v.so SWEEP_LW=2 SWEEP_NW=0 timeout 900 python profiles/loop_sweep.py c2 cold 1 5 2>&1 | grep -E "library default|one launch" > gpurun_out/r05_run12_$v.txt
echo $v; cat gpurun_out/r05_run12_$v.txt
done
