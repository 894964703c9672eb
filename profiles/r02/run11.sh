cd $GRAFT_REPO_ROOT
O=gpurun_out/r02k; mkdir -p $O
free -g | head -2; nproc
(SAGEICP_FORCE_BIG=1 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "correspondences or golden or c2_scaled or c5_scaled or c1_plumbing or edge_inputs or streaming" > $O/pytest_big.txt 2>&1); grep -E "passed|failed|Error|error|assert" $O/pytest_big.txt | tail -6
timeout 300 python profiles/knob_probe.py "" "SAGEICP_FORCE_BIG=1" 2>&1 | grep -v amdgpu
