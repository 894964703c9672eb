# k_fin's accumulator reduction on 256 threads instead of 1,024: phases, parity subset, c2 / stream / shards
timeout 600 python profiles/fin_phases.py c2 c4 2>&1 | grep -v "^$" | tail -12
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "c2 or c1 or rccl or direct or multi_device or chunked or reproduc" 2>&1 | tail -2
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline --steps 10 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c2 cold:', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"; done
timeout 300 python profiles/stream_probe.py 2>&1 | grep "per frame"
timeout 600 python profiles/shard_probe.py c2 cold 2>&1 | tail -4
