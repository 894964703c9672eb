# k_up_insert A/B: main (one fixed-size block per voxel) vs size classes on / off, kernel trace of the 40-frame stream
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
R=$PWD
kt() {  # tag dir env
  ( cd $2; env $3 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stream_kt19_$1 -o kt -- python profiles/stream_probe.py > $R/gpurun_out/stream_kt19_$1.txt 2>&1 )
  grep "per frame" gpurun_out/stream_kt19_$1.txt
  python - $1 <<'PY'
import csv, glob, sys
f = glob.glob('gpurun_out/stream_kt19_%s/**/kt_kernel_stats.csv' % sys.argv[1], recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in rows:
    if 'k_up_' in r['Name'] or 'k_far' in r['Name']:
        print('%-8s %-60s %6s calls  avg %8.2f us' % (sys.argv[1], r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3))
PY
}
kt main _main A=1
kt classes . A=1
kt oneclass . SAGEICP_SIZE_CLASSES=0
