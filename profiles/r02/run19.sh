# resident workgroups per CU limited by LDS padding: 3 / 4 / 5 / 6 / (8 = default build)
for o in res3 res4 res5 res6 occ8; do KNOB_CHILD="$o" KNOB_LIB=variants/$o.so python profiles/knob_probe.py; done
