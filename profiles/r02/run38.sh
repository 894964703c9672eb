cd $GRAFT_REPO_ROOT
O=gpurun_out/r02z; mkdir -p $O
(timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1); grep -E "passed|failed|Error|error|assert" $O/pytest.txt | tail -8
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -3
timeout 900 python profiles/knob_probe.py "" "" > $O/knob.txt 2>&1; cat $O/knob.txt
bash profiles/run_profiles.sh r02 > $O/prof.txt 2>&1; tail -30 $O/prof.txt
for wl in c1 c4 c5; do timeout 600 python bench.py --workload $wl --no-cpu-baseline > $O/bench_$wl.json 2> $O/bench_$wl.err; python -c "
import json,sys; d=json.load(open('$O/bench_$wl.json')); r=d['roofline']; print('$wl', d['value'],'fps', d['ms_per_step'],'ms', d['config']['iterations_per_frame'],'it; k_icp', r['avg_launch_us'],'us lanes', r['lanes_per_query'],'pairs frac', r['pairs_evaluated_frac'], 'cand/q', r['candidates_per_query'])"; done
timeout 600 python bench.py --params steady --no-cpu-baseline > $O/bench_c2_steady.json 2> $O/err; python -c "
import json; d=json.load(open('$O/bench_c2_steady.json')); print('c2 steady', d['value'], d['ms_per_step'], d['config']['iterations_per_frame'])"
timeout 600 python bench.py --workload c5 --params dense_nosem --no-cpu-baseline > $O/bench_c5_nosem.json 2> $O/err; python -c "
import json; d=json.load(open('$O/bench_c5_nosem.json')); print('c5 nosem', d['value'], d['ms_per_step'], d['config']['iterations_per_frame'])"
timeout 600 python profiles/stream_probe.py > $O/stream.txt 2>&1; tail -5 $O/stream.txt
timeout 300 python profiles/shard_probe.py > $O/shard.txt 2>&1; cat $O/shard.txt
