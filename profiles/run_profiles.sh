#!/bin/bash
# Collect the rocprofv3 evidence for one round.  Usage (on the GPU box, from the repo root):
#   bash profiles/run_profiles.sh <tag>        -> gpurun_out/prof_<tag>/...
# Every profiler invocation is wrapped in `timeout` (a failed counter config once hung
# rocprofv3's signal handler for 20 minutes).  PMC passes are separate runs and never mix with
# --kernel-trace/--stats, and stay within 4 TCC / 8 SQ counter slots.
TAG=${1:-r01}
R=$(pwd)
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $BENCH --steps 3 --warmup 1 > $OUT/bench_kt.json 2> $OUT/kt.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "k_nn" --output-format csv -d $OUT/pmc_fetch -o pmc -- $BENCH --steps 1 --warmup 0 > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-include-regex "k_nn" --output-format csv -d $OUT/pmc_tcc -o pmc -- $BENCH --steps 1 --warmup 0 > $OUT/bench_pmc_tcc.json 2> $OUT/pmc_tcc.err
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD --kernel-include-regex "k_nn" --output-format csv -d $OUT/pmc_sq -o pmc -- $BENCH --steps 1 --warmup 0 > $OUT/bench_pmc_sq.json 2> $OUT/pmc_sq.err
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS --kernel-include-regex "k_nn" --output-format csv -d $OUT/pmc_sq2 -o pmc -- $BENCH --steps 1 --warmup 0 > $OUT/bench_pmc_sq2.json 2> $OUT/pmc_sq2.err
cd $R
# keep only the small summaries (the per-dispatch CSVs of 555 launches are small too)
find $OUT -name "*.db" -delete
du -sh $OUT; find $OUT -type f | head -40
