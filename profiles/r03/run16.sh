# timeline of one k_icp launch and per-phase clocks on the round-3 code (instrumented build, -DSAGE_NN_TIMING)
export SPAN_LIB=$PWD/sage-icp_amd/_probe/libsageicp_tNN.so
timeout 600 python profiles/span_probe.py 1 20 100 > gpurun_out/span_r03.txt 2>&1; cat gpurun_out/span_r03.txt
timeout 600 python profiles/span_probe.py 20 100 --div 8 > gpurun_out/span_r03_15k.txt 2>&1; cat gpurun_out/span_r03_15k.txt
