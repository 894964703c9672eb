# survivors' voxel hashes computed on the device (off the replaying host thread): stream tests + probe
timeout 1500 python -m pytest tests/test_pipeline.py tests/test_kitti_io.py tests/test_robin_order.py tests/test_map_update_device.py -m gpu -x -q > gpurun_out/gputests_run15.txt 2>&1; grep -n "passed\|failed" gpurun_out/gputests_run15.txt; grep -n "Error\|assert" gpurun_out/gputests_run15.txt | head
timeout 300 python profiles/stream_probe.py 2>&1 | grep "per frame"
SAGEICP_DEBUG_ORDER=1 timeout 300 python profiles/stream_probe.py 2>&1 | grep "order level" | tail -3
timeout 300 python profiles/stream_probe.py 2>&1 | grep "per frame"
