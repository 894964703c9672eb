# compact candidate records (fp32 filter in front of the fp64 comparison)
python profiles/knob_probe.py "" "SAGEICP_NO_FILTER=1"
KNOB_CHILD="occ6" KNOB_LIB=variants/occ6.so python profiles/knob_probe.py
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -8
