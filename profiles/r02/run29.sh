# per-query streams (rows, frame, previous records) with the non-temporal hint vs plain
for v in base nt base nt; do SAGEICP_REGROUP=0 KNOB_CHILD="$v" KNOB_LIB=variants/$v.so python profiles/knob_probe.py; done
