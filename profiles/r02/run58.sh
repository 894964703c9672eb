# HIP runtime launch knobs: kernel arguments in device memory, fence scope at kernel boundaries
run() { env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline $WL 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   %.1f frames/s  %.3f ms' % (d['value'], d['ms_per_step']))"; }
for WL in "" "--workload c1"; do
echo "workload ${WL:-c2}"
for e in X=0 HIP_FORCE_DEV_KERNARG=0 HIP_FORCE_DEV_KERNARG=1 AMD_OPT_FLUSH=0 AMD_OPT_FLUSH=1 ROC_USE_FGS_KERNARG=0 DEBUG_HIP_KERNARG_COPY_OPT=0 ROC_SYSTEM_SCOPE_SIGNAL=0 X=0; do echo "  $e"; run $e; done
done > gpurun_out/runtime_knobs.txt 2>&1
cat gpurun_out/runtime_knobs.txt
