#!/bin/bash
# Collect the rocprofv3 evidence for one workload of one round.  Usage (on the GPU box, repo root):
#   bash profiles/run_profiles.sh <tag> [workload] [params]     -> gpurun_out/prof_<tag>_<workload>-<params>/...
# e.g.  bash profiles/run_profiles.sh r03            (c2 cold: the bench default)
#       bash profiles/run_profiles.sh r03 c4 steady
# Every profiler invocation is wrapped in `timeout`.  PMC passes are separate runs, never mixed
# with --kernel-trace/--stats, and stay within 4 TCC / 8 SQ counter slots (FETCH_SIZE alone takes 3).
TAG=${1:-r03}
WL=${2:-c2}
PRM=${3:-cold}
R=$(pwd)
OUT=$R/gpurun_out/prof_${TAG}_${WL}-${PRM}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --workload $WL --params $PRM --no-cpu-baseline"
K="k_icp<|k_fin|k_rows|k_loop<"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $BENCH --steps 3 --warmup 1 > $OUT/bench_kt.json 2> $OUT/kt.err
# The counter passes run one kernel at a time (rocprofv3 --pmc serialises dispatches), which the one-launch loop —
# a grid and its solving wave, two kernels that talk to each other — does not survive: they use the counter-
# collection twin of the library (sage-icp_amd/_probe/libsageicp_ingrid.so, -DSAGE_LOOP_INGRID: the solving wave
# is one more workgroup of the grid; same search code, same bytes and instructions, not the same time).
if [ -f $R/sage-icp_amd/_probe/libsageicp_ingrid.so ]; then export SAGEICP_VARIANT_LIB=$R/sage-icp_amd/_probe/libsageicp_ingrid.so; fi
# ... and the chained launches of the frames beyond the LDS (k_icp beside the resident solving wave: two kernels that talk
# to each other as well) run with k_fin between them (SAGEICP_CHAIN=0): the same k_icp, the same bytes and instructions.
export SAGEICP_CHAIN=0
PM="$BENCH --steps 1 --warmup 0 --no-profile-events"
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$K" --output-format csv -d $OUT/pmc_fetch -o pmc -- $PM > /dev/null 2> $OUT/pmc_fetch.err
timeout 400 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-include-regex "$K" --output-format csv -d $OUT/pmc_tcc -o pmc -- $PM > /dev/null 2> $OUT/pmc_tcc.err
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-include-regex "$K" --output-format csv -d $OUT/pmc_sq -o pmc -- $PM > /dev/null 2> $OUT/pmc_sq.err
timeout 400 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-include-regex "$K" --output-format csv -d $OUT/pmc_sq2 -o pmc -- $PM > /dev/null 2> $OUT/pmc_sq2.err
timeout 400 rocprofv3 --pmc TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS --kernel-include-regex "$K" --output-format csv -d $OUT/pmc_mem -o pmc -- $PM > /dev/null 2> $OUT/pmc_mem.err
unset SAGEICP_VARIANT_LIB
unset SAGEICP_CHAIN
cd $R
# the un-profiled line of the same workload (with the CPU baseline and the parity block)
timeout 900 python bench.py --workload $WL --params $PRM > $OUT/bench_default.json 2> $OUT/bench_default.err
find $OUT -name "*.db" -delete
python profiles/summarize.py $OUT > $OUT/summary.md 2>&1
cat $OUT/summary.md
