# probe (wrong answers): only 16 of the 32 bytes of every scanned point are loaded
SAGEICP_REGROUP=0 KNOB_CHILD="halfload" KNOB_LIB=variants/half.so python profiles/knob_probe.py
SAGEICP_REGROUP=0 KNOB_CHILD="full" python profiles/knob_probe.py
