timeout 900 python -m pytest tests/test_map_update_device.py -m gpu -x -q 2>&1 | tail -5
timeout 300 python profiles/pointcloud_probe.py 2>&1 | tail -4
