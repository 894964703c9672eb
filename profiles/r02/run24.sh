# work regrouping inside chunks of 1024 queries after iteration 2
python profiles/knob_probe.py "SAGEICP_REGROUP=0" "SAGEICP_REGROUP=1" "SAGEICP_REGROUP=1 SAGEICP_REGROUP_AT=1" "SAGEICP_REGROUP=1 SAGEICP_REGROUP_AT=4"
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error" | tail -5
