timeout 900 python -m pytest tests/test_map_update_device.py tests/test_pipeline.py -m gpu -x -q 2>&1 | tail -2
sed -i 's/^kt main _main A=1$//; s/^kt oneclass . SAGEICP_SIZE_CLASSES=0$//' profiles/r03/run19.sh
bash profiles/r03/run19.sh
timeout 300 python profiles/stream_probe.py 2>&1 | grep "per frame"
