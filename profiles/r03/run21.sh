# where k_up_insert's time goes: the phase probe build (-DSAGE_UP_TIMING), classes on / off
cp sage-icp_amd/libsageicp_hip.so /tmp/keep.so; cp sage-icp_amd/libsageicp_hip.uptiming.so sage-icp_amd/libsageicp_hip.so
timeout 300 python profiles/stream_probe.py > gpurun_out/up_timing_classes.txt 2>&1; grep "k_up_insert phases" gpurun_out/up_timing_classes.txt | sed -n '2p;10p;20p;38p'
SAGEICP_SIZE_CLASSES=0 timeout 300 python profiles/stream_probe.py > gpurun_out/up_timing_oneclass.txt 2>&1; grep "k_up_insert phases" gpurun_out/up_timing_oneclass.txt | sed -n '2p;10p;20p;38p'
cp /tmp/keep.so sage-icp_amd/libsageicp_hip.so
