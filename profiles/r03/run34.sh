# compile-time launch shape of k_icp re-measured on the final code: XCD stripe (8 workgroups) and waves per workgroup (4)
cp sage-icp_amd/libsageicp_hip.so /tmp/keep.so
one() {
  for a in "c2 cold" "c5 dense" "c4 steady"; do set -- $a
    timeout 600 python bench.py --workload $1 --params $2 --no-cpu-baseline --steps 8 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('  $1 $2:', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
  done
  timeout 300 python profiles/stream_probe.py 2>&1 | grep "per frame"
}
echo "default (stripe 8, 4 waves)"; one
for v in s4 s16 s32 w2 w8; do echo "variant $v"; cp sage-icp_amd/_probe/libsageicp_$v.so sage-icp_amd/libsageicp_hip.so; one; done
cp /tmp/keep.so sage-icp_amd/libsageicp_hip.so
