# round 6, run 11: GPU test suite with wave priorities + dealt first units as defaults; bench c2 / c1
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 > gpurun_out/r06/run11_tests.txt
cat gpurun_out/r06/run11_tests.txt
for w in c2 c1; do
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --workload $w 2>/dev/null | tail -1 > gpurun_out/r06/run11_bench_$w.json
done
python - <<'PY'
import json
for w in ("c2","c1"):
    d=json.loads(open("gpurun_out/r06/run11_bench_%s.json"%w).read())
    print(w, "%.1f frames/s %.3f ms  %.2f us/iteration host-entry %s" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], d.get("ms_per_step_host_entry")))
PY
