# lanes per query on the instruction-bound workloads: c4 steady / cold and c2 with SAGEICP_LW = 1 (2 lanes), 2 (4 lanes, the choice), 3 (8 lanes)
for a in "c4 steady" "c4 cold" "c2 cold" "c5 dense"; do set -- $a
 for lw in 1 2 3; do
  SAGEICP_LW=$lw timeout 600 python bench.py --workload $1 --params $2 --no-cpu-baseline --steps 8 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 $2 lw=$lw:', d['value'], d['ms_per_step'], d['config']['iterations_per_frame'], d['roofline']['avg_launch_us'])"
 done
done
