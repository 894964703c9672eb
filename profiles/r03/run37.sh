cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
kt() {
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stream_kt37_$1 -o kt -- python profiles/stream_probe.py > gpurun_out/stream_kt37_$1.txt 2>&1
  python - $1 <<'PY'
import csv, glob, sys
f = glob.glob('gpurun_out/stream_kt37_%s/**/kt_kernel_stats.csv' % sys.argv[1], recursive=True)
for r in csv.DictReader(open(f[0])):
    if 'k_up_' in r['Name']: print('%-6s %-50s avg %8.2f us' % (sys.argv[1], r['Name'][30:80], float(r['AverageNs'])/1e3))
PY
}
cp sage-icp_amd/libsageicp_hip.so /tmp/keep.so
kt new
cp sage-icp_amd/_probe/libsageicp_prev.so sage-icp_amd/libsageicp_hip.so; kt prev
cp /tmp/keep.so sage-icp_amd/libsageicp_hip.so; kt new2
