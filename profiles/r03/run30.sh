for a in "c5 dense 0" "c5 dense 1" "c5 dense_nosem 0" "c5 dense_nosem 1" "c5 dense_nosem 2" "c4 steady 0" "c1 cold 1" "c1 cold 2" "c1 cold 0"; do set -- $a
  SAGEICP_LW=$3 timeout 600 python bench.py --workload $1 --params $2 --no-cpu-baseline --steps 8 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 $2 lw=$3:', d['value'], d['ms_per_step'], d['config']['iterations_per_frame'], d['roofline']['avg_launch_us'])"
done
# c5 at scales (fewer queries against the same kind of map): where does 2 lanes stop paying?
for sc in 0.5 0.25; do for lw in 1 2; do
  SAGEICP_LW=$lw timeout 600 python bench.py --workload c5 --params dense --scale $sc --no-cpu-baseline --steps 8 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c5 scale $sc lw=$lw:', d['value'], d['ms_per_step'], d['config']['iterations_per_frame'], d['roofline']['avg_launch_us'], d['config'].get('n_queries'))"
done; done
