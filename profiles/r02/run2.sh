cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
(timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r02b/pytest.txt 2>&1); grep -E "passed|failed|Error|error|assert" gpurun_out/r02b/pytest.txt | tail -12
