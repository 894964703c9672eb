cd $GRAFT_REPO_ROOT
O=gpurun_out/r02h; mkdir -p $O
(timeout 1500 python -m pytest tests/test_kitti_io.py tests/test_shim_compile.py -m gpu -q -x -s > $O/pytest.txt 2>&1); grep -E "passed|failed|Error|error|assert|c3 stream" $O/pytest.txt | tail -8
timeout 300 python profiles/knob_probe.py "SAGEICP_LW=3" "SAGEICP_LW=4" 2>&1 | grep -v amdgpu > $O/knob.txt; cat $O/knob.txt
for lw in 3 4; do SAGEICP_LW=$lw timeout 300 python bench.py --workload c1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c1 lw $lw', d['value'], d['ms_per_step'])"; done
for lw in 1 2; do SAGEICP_LW=$lw timeout 300 python bench.py --workload c4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c4 lw $lw', d['value'], d['ms_per_step'])"; done
