# c4 (500k-pt scan vs 10M-pt map): bytes from beyond the L2s per k_icp launch, and the kernel's duration
R=$(pwd); OUT=$R/gpurun_out/pmc_c4; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --workload c4 --no-cpu-baseline --no-profile-events"
K="k_icp|k_fin"
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $BENCH --steps 2 --warmup 1 > $OUT/bench_kt.json 2> $OUT/kt.err
timeout 500 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$K" --output-format csv -d $OUT/pmc_fetch -o pmc -- $BENCH --steps 1 --warmup 0 > /dev/null 2> $OUT/pmc_fetch.err
timeout 500 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-include-regex "$K" --output-format csv -d $OUT/pmc_tcc -o pmc -- $BENCH --steps 1 --warmup 0 > /dev/null 2> $OUT/pmc_tcc.err
cd $R
find $OUT -name "*.db" -delete
{ echo "c4 k_icp, rocprofv3 (kernel-trace --stats; --pmc FETCH_SIZE; --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum):";
  grep -h "k_icp\|k_fin" $(find $OUT/kt -name "*kernel_stats.csv") | cut -c1-160;
  python profiles/pmc_summary.py $OUT/pmc_fetch k_icp; python profiles/pmc_summary.py $OUT/pmc_tcc k_icp; } > gpurun_out/pmc_c4.txt 2>&1
cat gpurun_out/pmc_c4.txt; tail -2 $OUT/*.err
rm -rf $OUT/kt $OUT/pmc_fetch $OUT/pmc_tcc
