cd $GRAFT_REPO_ROOT
O=gpurun_out/r02m; mkdir -p $O
(timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1); grep -E "passed|failed|Error|error|assert" $O/pytest.txt | tail -8
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -3
