cd $GRAFT_REPO_ROOT
R=$(pwd); OUT=$R/gpurun_out/r02e; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for d in 1200 120 30; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt$d -o kt -- python $R/profiles/shard_run.py $d 3 > $OUT/run$d.txt 2> $OUT/err$d.txt
cat $OUT/run$d.txt | tail -1
python - <<PY
import csv
for r in list(csv.DictReader(open("$OUT/kt$d/kt_kernel_stats.csv")))[:2]:
    print("  %-60s calls %6s avg %8.2f us  min %8.2f max %8.2f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
done
find $OUT -name "*.db" -delete
