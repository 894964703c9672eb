mkdir -p gpurun_out/r05_kt_c1
R=$(pwd); cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05_kt_c1 -o kt -- python $R/bench.py --workload c1 --no-cpu-baseline --steps 20 --warmup 3 --no-profile-events > $R/gpurun_out/r05_kt_c1/bench.json 2> $R/gpurun_out/r05_kt_c1/err.txt
cd $R
python - <<'PY'
import csv,glob,json
p=glob.glob('gpurun_out/r05_kt_c1/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(p)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last frame: find the last k_loop< launch and print the kernels around it
idx=[i for i,r in enumerate(rows) if 'k_loop<' in r['Kernel_Name']]
i=idx[-2]
t0=int(rows[i-12]['Start_Timestamp'])
for r in rows[i-12:i+6]:
    print("%9.1f us  +%7.1f us  %s" % ((int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, r['Kernel_Name'].split('(')[0][:70]))
d=json.load(open('gpurun_out/r05_kt_c1/bench.json')); print(d['value'], d['ms_per_step'])
PY
find gpurun_out/r05_kt_c1 -name "*.db" -delete
