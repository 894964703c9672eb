# probe (wrong answers): scans truncated after 48 / 96 / 144 points per query — what do the heaviest chains cost?
for t in 48 96 144; do SAGEICP_REGROUP=0 KNOB_CHILD="trunc$t" KNOB_LIB=variants/trunc$t.so python profiles/knob_probe.py; done
SAGEICP_REGROUP=0 KNOB_CHILD="full" python profiles/knob_probe.py
