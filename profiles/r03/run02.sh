# what would hiding k_fin under k_icp's prologue cost?  probe build: every wave waits D ticks (100 MHz) after
# its start for "the pose", its prologue loads (row, frame point, previous answer) already in flight
export KNOB_LIB=$PWD/sage-icp_amd/_probe/libsageicp_delay.so
timeout 900 python profiles/knob_probe.py "SAGEICP_DBG_DELAY=0" "SAGEICP_DBG_DELAY=200" "SAGEICP_DBG_DELAY=400" "SAGEICP_DBG_DELAY=600" "SAGEICP_DBG_DELAY=800" "SAGEICP_DBG_DELAY=1000" "SAGEICP_DBG_DELAY=0" > gpurun_out/delay_probe.txt 2>&1
cat gpurun_out/delay_probe.txt
