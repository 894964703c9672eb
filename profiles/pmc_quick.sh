R=$(pwd); OUT=$R/gpurun_out/pmc_quick; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 120 rocprofv3 -L > $OUT/counters_list.txt 2>&1
B="python $R/bench.py --no-cpu-baseline --steps 1 --warmup 0"
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-include-regex "k_nn" --output-format csv -d $OUT/a -o pmc -- $B > /dev/null 2> $OUT/a.err
timeout 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU --kernel-include-regex "k_nn" --output-format csv -d $OUT/b -o pmc -- $B > /dev/null 2> $OUT/b.err
timeout 200 rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-include-regex "k_nn" --output-format csv -d $OUT/c -o pmc -- $B > /dev/null 2> $OUT/c.err
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $B > /dev/null 2> $OUT/kt.err
cd $R; find $OUT -name "*.db" -delete; ls $OUT/*
