for a in 1 0; do echo "== SAGEICP_POINTCLOUD_AHEAD=$a"; SAGEICP_POINTCLOUD_AHEAD=$a STREAM_LOCALMAP=1 timeout 300 python profiles/stream_probe.py 2>&1 | grep -E "per frame|LocalMap\(\) per"; done
timeout 900 python -m pytest tests/test_pipeline.py tests/test_map_update_device.py tests/test_reference_order_map.py tests/test_kitti_io.py -q -m gpu 2>&1 | tail -4
