cd $GRAFT_REPO_ROOT
O=gpurun_out/r02i; mkdir -p $O
(timeout 1500 python -m pytest tests/test_kitti_io.py tests/test_pipeline.py -m gpu -q -x -s > $O/pytest.txt 2>&1); grep -E "passed|failed|Error|error|assert|c3 stream" $O/pytest.txt | tail -8
OMP_NUM_THREADS=32 timeout 1500 python profiles/robin_order_probe.py 200 30000 2>&1 | grep -v amdgpu > $O/robin.txt; cat $O/robin.txt
timeout 600 python profiles/stream_probe.py > $O/stream.txt 2>&1; tail -4 $O/stream.txt
