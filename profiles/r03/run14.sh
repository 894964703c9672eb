# PC sampling of k_icp (beta): which configurations exist, then one short sampled run of c2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pcs; mkdir -p $OUT
timeout 60 rocprofv3 -L 2>&1 | grep -i -B2 -A12 "sampling" | head -60 > $OUT/avail.txt; cat $OUT/avail.txt
BENCH="python $R/bench.py --no-cpu-baseline --no-profile-events --steps 2 --warmup 1"
timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit time --pc-sampling-method host_trap --pc-sampling-interval 1 --output-format csv -d $OUT/ht -o pcs -- $BENCH > $OUT/ht.json 2> $OUT/ht.err; echo "host_trap rc=$?"; tail -3 $OUT/ht.err; ls -la $OUT/ht 2>/dev/null | head
find $OUT -name "*.csv" | head; for f in $(find $OUT -name "*pc_sampling*.csv" | head -2); do echo $f; head -5 $f; wc -l $f; done
find $OUT -name "*.db" -delete
