cd $GRAFT_REPO_ROOT
O=gpurun_out/r02j; mkdir -p $O
(timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "multi_device or rccl_refuses or direct_exchange" > $O/pytest.txt 2>&1); grep -E "passed|failed|Error|error|assert" $O/pytest.txt | tail -12
