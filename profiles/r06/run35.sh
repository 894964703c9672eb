mkdir -p gpurun_out/r06
for w in c4 c5 c2 c1; do
timeout 900 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --workload $w 2>/dev/null | tail -1 > gpurun_out/r06/run35_bench_$w.json
done
python - <<'PY'
import json
for w in ("c4","c5","c2","c1"):
    d=json.loads(open("gpurun_out/r06/run35_bench_%s.json"%w).read())
    print(w, "%.1f frames/s %.3f ms  %.2f us/launch  host-entry %s | %s | parity %s" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], d.get("ms_per_step_host_entry"), d["roofline"]["loop_form"][:60], d.get("parity")))
PY
