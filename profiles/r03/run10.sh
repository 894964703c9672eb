# what does a pass cost on warm L2s (no kernel boundary before it)?  probe build: the whole body R more times per launch
export KNOB_LIB=$PWD/sage-icp_amd/_probe/libsageicp_delay.so
timeout 900 python profiles/knob_probe.py "SAGEICP_DBG_REPEAT=0" "SAGEICP_DBG_REPEAT=1" "SAGEICP_DBG_REPEAT=2" "SAGEICP_DBG_REPEAT=4" > gpurun_out/repeat_probe.txt 2>&1
cat gpurun_out/repeat_probe.txt
