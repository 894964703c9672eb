cd /tmp && export TMPDIR=/tmp
R=/root/repo
for wl in c1 c2; do
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/sortkt_$wl -o kt -- python $R/bench.py --workload $wl --params cold --no-cpu-baseline --steps 6 --warmup 2 > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob('$R/gpurun_out/sortkt_$wl/**/kt_kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k_loop<' in r['Kernel_Name']]
i0,i1=idx[-2],idx[-1]
t0=int(rows[i0]['End_Timestamp'])
print('$wl')
for r in rows[i0:i1+1]:
    print('%9.2f %9.2f  %s' % ((int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, r['Kernel_Name'][:70]))
PY
done
