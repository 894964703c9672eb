# round 6, run 20: GPU test suite (heaviest-first k_icp, two-step filter, in-band overflow, UTM frame test, c4 fixture); bench c2 / c4
mkdir -p gpurun_out/r06
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 > gpurun_out/r06/run20_tests.txt
cat gpurun_out/r06/run20_tests.txt
for w in c2 c4; do
timeout 900 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --workload $w 2>/dev/null | tail -1 > gpurun_out/r06/run20_bench_$w.json
done
python - <<'PY'
import json
for w in ("c2","c4"):
    d=json.loads(open("gpurun_out/r06/run20_bench_%s.json"%w).read())
    print(w, "%.1f frames/s %.3f ms  %.2f us/iteration host-entry %s" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], d.get("ms_per_step_host_entry")))
PY
