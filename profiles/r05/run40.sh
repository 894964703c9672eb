# bench lines of this tree with its own counters in place (profiles/icp_counters.json)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_c2_cold.json 2> gpurun_out/r05_bench_c2_cold.err
for w in "c1 cold" "c2 steady" "c4 steady" "c4 cold" "c5 dense" "c5 dense_nosem"; do timeout 900 python bench.py --workload ${w% *} --params ${w#* } --no-cpu-baseline > gpurun_out/r05_bench_${w% *}_${w#* }.json 2>/dev/null; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05_bench_c*.json')):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,e); continue
    r=d['roofline']; print(f.split('/')[-1], d['value'], d['ms_per_step'], r.get('avg_launch_us'), 'frac', r.get('frac'), r.get('counters_fresh'), r.get('traffic'), (r.get('frac_basis') or '')[:60])
PY
