# compact candidates with parked fetches (two per lane, flushed together)
python profiles/knob_probe.py "" ""
KNOB_CHILD="occ6" KNOB_LIB=variants/occ6.so python profiles/knob_probe.py
SPAN_LIB=variants/tNN.so python profiles/span_probe.py 1 20 100 2>&1 | grep -E "iteration|pair steps|lifetime us"
KNOB_LIB=variants/occ6.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5
