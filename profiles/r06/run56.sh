cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pcs; mkdir -p $O
(rocprofv3-avail list --pc-sampling 2>&1 || rocprofv3-avail pc-sample-config 2>&1 || rocprofv3-avail --help 2>&1) | head -40
for cfg in "stochastic cycles 1048576" "stochastic cycles 65536" "host_trap time 1" "host_trap time 10000"; do
  set -- $cfg
  timeout 200 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $1 --pc-sampling-unit $2 --pc-sampling-interval $3 --kernel-trace --output-format csv -d $O/raw_$1_$3 -o pcs -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile-events > $O/bench_$1_$3.json 2> $O/err_$1_$3.txt
  echo "cfg=$cfg rc=$?"; grep -i "not supported\|error" $O/err_$1_$3.txt | head -2
  if ls $O/raw_$1_$3/*/*pc_sampling*.csv > /dev/null 2>&1; then
    ls -la $O/raw_$1_$3/*/ | head; python $R/profiles/pc_hist.py $O/raw_$1_$3 k_loop > $O/pc_hist_$1_$3.txt 2>&1; head -12 $O/pc_hist_$1_$3.txt
  fi
  rm -rf $O/raw_$1_$3
done
