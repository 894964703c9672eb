# k_fin: loop state fetched with the partials (an idle thread parks it in LDS)
python profiles/knob_probe.py "" ""
PHASE_LIB=variants/tGN.so PHASE_KIND=GN python profiles/phase_probe.py 1 8
python bench.py --workload c1 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c1', d['value'], 'frames/s', d['ms_per_step'], 'ms')"
python profiles/stream_probe.py 2>&1 | grep -E "per frame"
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
