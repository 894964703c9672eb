timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2-cold %.1f frames/s  %.3f ms  %.2f us/iteration' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_us']))"
rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk" | head -3
