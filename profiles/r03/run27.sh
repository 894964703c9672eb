for t in 4 8 2 0; do echo "THREADS=$t"; SAGEICP_TOUCH_THREADS=$t timeout 300 python profiles/pointcloud_probe.py 2>&1 | tail -3 | head -2; done
STREAM_LOCALMAP=1 timeout 300 python profiles/stream_probe.py 2>&1 | grep "per frame\|LocalMap() per"
timeout 600 python -m pytest tests/test_map_update_device.py tests/test_pipeline.py -m gpu -x -q 2>&1 | tail -2
