# compact candidates: the candidate already held is not looked at twice
python profiles/knob_probe.py "" ""
KNOB_CHILD="occ6" KNOB_LIB=variants/occ6.so python profiles/knob_probe.py
