# round-4 evidence on the final device code (flat-order scan, lane rules re-tuned for it); the counters are tied to the sources by hash
mkdir -p gpurun_out
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r04_gputests_final.txt
cat gpurun_out/r04_gputests_final.txt
bash profiles/run_profiles.sh r04 c2 cold > /dev/null 2>&1
bash profiles/run_profiles.sh r04 c5 dense > /dev/null 2>&1
bash profiles/run_profiles.sh r04 c4 steady > /dev/null 2>&1
for wl in "c1 cold" "c2 steady" "c4 cold" "c5 dense_nosem"; do set -- $wl; timeout 900 python bench.py --workload $1 --params $2 --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/r04_bench_$1_$2.json; done
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep '^{' > gpurun_out/r04_bench_driver_cmd.json
python -c "
import json,glob
for p in sorted(glob.glob('gpurun_out/r04_bench_*.json'))+sorted(glob.glob('gpurun_out/prof_r04_*/bench_default.json')):
    try:
        d=json.load(open(p)); print(p, d['value'], 'frames/s', d['ms_per_step'], 'ms', d['config']['iterations_per_frame'], 'it')
    except Exception as e: print(p, e)
"
head -30 gpurun_out/prof_r04_c2-cold/summary.md
