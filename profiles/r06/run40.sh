# what the frame sort's launches cost c1: the same frame with the sort replaced by one copy (the frame unsorted: the loop itself gets slower)
for v in 0 1; do
  SAGEICP_DEBUG_NO_SORT=$v timeout 300 python bench.py --workload c1 --params cold --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
it=r['launches_timed']/d['steps']
print('no_sort=$v ms/frame %.4f  k_loop %.2f us/it x %.0f it = %.1f us  fixed %.1f us  host entry %.4f' % (d['ms_per_step'], r['avg_launch_us'], it, r['avg_launch_us']*it, d['ms_per_step']*1e3-r['avg_launch_us']*it, d['ms_per_step_host_entry']))"
done
