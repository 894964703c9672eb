# round 6, run 19: granularity of the heaviest-first order: stripes of 8 (product), 4, 2, 1 workgroups
mkdir -p gpurun_out/r06
export SAGEICP_LPT_MAX=1
for k in 8 4 2 1; do
  if [ $k = 8 ]; then lib=""; else lib=sage-icp_amd/_probe/libsageicp_stripe$k.so; fi
  echo "== stripes of $k workgroups"
  KNOB_LIB=$lib KNOB_REPS=2 timeout 900 python profiles/knob_ab.py "c4:steady:1:4 c5:dense:1:10" "SAGEICP_LPT=0" "SAGEICP_LPT=1" 2>&1
done | tee gpurun_out/r06/lpt_stripe_ab.txt
