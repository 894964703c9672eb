# campaign: 30 x the usual number of random examples of every hypothesis test that runs on the GPU
mkdir -p gpurun_out/r06
SAGE_TEST_EXAMPLES=30 timeout 3300 python -m pytest tests/test_reference_order_map.py tests/test_map_update_device.py tests/test_gpu_parity.py -m gpu -q -k "random_update_sequences or device_update_property or get_correspondences_property or register_frame_property or clone" --durations=6 2>&1 | tail -25 | tee gpurun_out/r06/campaign.txt
