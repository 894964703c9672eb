cd $GRAFT_REPO_ROOT
O=gpurun_out/r02l; mkdir -p $O
(timeout 700 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "ten_million" --durations=3 > $O/pytest.txt 2>&1); grep -E "passed|failed|Error|error|assert" $O/pytest.txt | tail -8
