# the driver's commands on the round's final tree
timeout 600 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/bench_final.json') if l.startswith('{')][-1])
r=d['roofline']
print(d['metric'], d['value'], d['unit'], d['ms_per_step'], 'frac', r['frac'], 'fresh', r['counters_fresh'], 'achieved', r['achieved'], 'traffic', r['traffic'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], 'parity', d['parity']['ok'])
PY
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
