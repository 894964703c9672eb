# round 6, run 23: reference-order maps updated on the device
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_reference_order_map.py tests/test_map_update_device.py tests/test_pipeline.py tests/test_kitti_io.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert|frame" | tail -12
