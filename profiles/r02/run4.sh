cd $GRAFT_REPO_ROOT
R=$(pwd); OUT=$R/gpurun_out/r02d; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for d in 8 1; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt$d -o kt -- python $R/profiles/shard_run.py $d 3 > $OUT/run$d.txt 2> $OUT/err$d.txt
cat $OUT/run$d.txt | tail -1
python - <<PY
import csv
for r in list(csv.DictReader(open("$OUT/kt$d/kt_kernel_stats.csv")))[:6]:
    print("  %-60s calls %6s avg %8.2f us  %5s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
done
find $OUT -name "*.db" -delete
