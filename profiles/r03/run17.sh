# PROBE: the heaviest queries (neighbourhood > factor x mean points) searched by a second launch with more lanes per query
for f in 0 1.5 2.0 3.0; do
  for plus in 2 1; do
    [ "$f" = "0" ] && [ "$plus" = "1" ] && continue
    echo "== factor $f, heavy lanes = light lanes << $plus"
    SAGEICP_HEAVY_PROBE=$f SAGEICP_HEAVY_LW_PLUS=$plus timeout 600 python bench.py --no-cpu-baseline --steps 6 --warmup 2 2> gpurun_out/hp.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c2', d['value'], d['ms_per_step'], d['config']['iterations_per_frame'], d['roofline']['avg_launch_us'])"
    grep "heavy probe" gpurun_out/hp.err | tail -1
  done
done 2>&1 | tee gpurun_out/heavy_probe.txt
echo "== 15k shard" | tee -a gpurun_out/heavy_probe.txt
for f in 0 1.5 2.5; do
  SAGEICP_HEAVY_PROBE=$f SAGEICP_HEAVY_LW_PLUS=1 timeout 600 python profiles/knob_probe.py "" 2> gpurun_out/hp.err | tee -a gpurun_out/heavy_probe.txt
  grep "heavy probe" gpurun_out/hp.err | tail -2 | tee -a gpurun_out/heavy_probe.txt
done
