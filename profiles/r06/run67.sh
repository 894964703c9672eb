# campaign, larger: 300 x (12,000 sequences) for the reference-order device update, 100 x for the search / registration / device-update properties
mkdir -p gpurun_out/r06
(SAGE_TEST_EXAMPLES=300 timeout 1500 python -m pytest tests/test_reference_order_map.py -m gpu -q -k "random_update_sequences" 2>&1 | tail -4
SAGE_TEST_EXAMPLES=100 timeout 2400 python -m pytest tests/test_map_update_device.py tests/test_gpu_parity.py -m gpu -q -k "device_update_property or get_correspondences_property or register_frame_property" --durations=4 2>&1 | tail -10) | tee gpurun_out/r06/campaign2.txt
