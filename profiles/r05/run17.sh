mkdir -p gpurun_out/r05_pmc_try
R=$(pwd); cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-include-regex "k_loop<|k_icp<" --output-format csv -d $R/gpurun_out/r05_pmc_try -o pmc -- python $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 --no-profile-events > $R/gpurun_out/r05_pmc_try/bench.json 2> $R/gpurun_out/r05_pmc_try/err.txt
cd $R
tail -3 gpurun_out/r05_pmc_try/err.txt
python - <<'PY'
import csv,collections,glob,json
p=glob.glob('gpurun_out/r05_pmc_try/**/pmc_counter_collection.csv',recursive=True)
acc=collections.defaultdict(list)
for r in csv.DictReader(open(p[0])):
    acc[(r['Kernel_Name'].split('(')[0][:50], r['Counter_Name'])].append((float(r['Counter_Value']), float(r['End_Timestamp'])-float(r['Start_Timestamp'])))
for k,v in sorted(acc.items()):
    print(k, len(v), 'mean %.4g'%(sum(x[0] for x in v)/len(v)), 'dur us %.1f'%(sum(x[1] for x in v)/len(v)/1e3))
d=json.load(open('gpurun_out/r05_pmc_try/bench.json')); print(d['value'], d['roofline']['loop_form'])
PY
find gpurun_out/r05_pmc_try -name "*.db" -delete
