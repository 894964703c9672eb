# occupancy variants of the slimmer scan state (no offset tracking): 6 / 7 / 8 waves per SIMD
for o in 6 7 8 7 8; do KNOB_CHILD="occ$o" KNOB_LIB=variants/occ$o.so python profiles/knob_probe.py; done
KNOB_CHILD="occ8" KNOB_LIB=variants/occ8.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
