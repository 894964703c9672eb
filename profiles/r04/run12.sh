# where does k_skip's time go?  kernel trace + SQ counters, c2 cold, against k_icp
R=$(pwd); OUT=$R/gpurun_out/prof_r04_skip; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for m in 0 1; do
  export SAGEICP_SKIP=$m SAGEICP_SKIP_MARGIN_MM=20 SAGEICP_LOOP=0
  B="python $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 --no-profile-events"
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt$m -o kt -- $B > /dev/null 2> $OUT/kt$m.err
  timeout 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-include-regex "k_icp|k_skip" --output-format csv -d $OUT/sq$m -o pmc -- $B > /dev/null 2> $OUT/sq$m.err
  timeout 400 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS --kernel-include-regex "k_icp|k_skip" --output-format csv -d $OUT/sq2$m -o pmc -- $B > /dev/null 2> $OUT/sq2$m.err
  timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "k_icp|k_skip" --output-format csv -d $OUT/f$m -o pmc -- $B > /dev/null 2> $OUT/f$m.err
done
cd $R
python - <<'PY'
import csv, glob, collections
for m in (0, 1):
    print("== SAGEICP_SKIP=%d" % m)
    for p in glob.glob("gpurun_out/prof_r04_skip/kt%d/**/*kernel_stats.csv" % m, recursive=True):
        for r in list(csv.DictReader(open(p)))[:4]:
            print("   ", r["Name"][:60], r["Calls"], "calls", "%.2f us avg" % (float(r["AverageNs"]) / 1e3), r["Percentage"], "%")
    for sub in ("sq", "sq2", "f"):
        for p in glob.glob("gpurun_out/prof_r04_skip/%s%d/**/*counter_collection.csv" % (sub, m), recursive=True):
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(p)):
                if float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) < 6000: continue
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            print("    " + "  ".join("%s %.4g" % (k, sum(v) / len(v)) for k, v in sorted(acc.items())))
PY
find gpurun_out/prof_r04_skip -name "*.db" -delete; find gpurun_out/prof_r04_skip -name "*trace.csv" -delete; find gpurun_out/prof_r04_skip -name "*counter_collection.csv" -delete
