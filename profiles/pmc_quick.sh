#!/bin/bash
# quick counter passes for k_nn / k_gn (one registration each); usage: bash profiles/pmc_quick.sh <tag>
TAG=${1:-quick}
R=$(pwd)
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --no-cpu-baseline --no-profile-events"
K="k_icp|k_fin|k_rows"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $BENCH --steps 3 --warmup 1 > $OUT/bench_kt.json 2> $OUT/kt.err
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-include-regex "$K" --output-format csv -d $OUT/pmc_sq -o pmc -- $BENCH --steps 1 --warmup 0 > /dev/null 2> $OUT/pmc_sq.err
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_INSTS_LDS --kernel-include-regex "$K" --output-format csv -d $OUT/pmc_sq2 -o pmc -- $BENCH --steps 1 --warmup 0 > /dev/null 2> $OUT/pmc_sq2.err
timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS --kernel-include-regex "$K" --output-format csv -d $OUT/pmc_mem -o pmc -- $BENCH --steps 1 --warmup 0 > /dev/null 2> $OUT/pmc_mem.err
cd $R
find $OUT -name "*.db" -delete
python profiles/summarize.py $OUT > $OUT/summary.md 2>&1
cat $OUT/summary.md
tail -2 $OUT/*.err
