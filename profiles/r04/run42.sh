#!/bin/bash
# instructions a wave of k_loop issues per iteration (15k-query shard, 16 and 8 lanes per query)
mkdir -p gpurun_out/pmc_loop
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lw in 4 3; do
  ( cd $R && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -d $R/gpurun_out/pmc_loop/lw$lw -o out --output-format csv -- python profiles/loop_pmc_run.py 8 $lw ) > $R/gpurun_out/pmc_loop/log_lw$lw.txt 2>&1
  tail -2 $R/gpurun_out/pmc_loop/log_lw$lw.txt
done
cd $R
python - <<'PY'
import csv, glob, collections
for lw in (4, 3):
    files = glob.glob('gpurun_out/pmc_loop/lw%d/**/*counter_collection.csv' % lw, recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:40]
            agg[k][r['Counter_Name']] += float(r['Counter_Value'])
            if r['Counter_Name'] == 'SQ_WAVES': cnt[k] += 1
    for k, v in agg.items():
        if 'k_loop' in k:
            print('lw', lw, k, 'launches', cnt[k], {c: round(x / max(cnt[k], 1)) for c, x in v.items()})
PY
