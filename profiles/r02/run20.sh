# what a k_icp pass costs with warm L2s: the body repeated 2x / 3x inside one launch
for o in occ8 reps2 reps3; do KNOB_CHILD="$o" KNOB_LIB=variants/$o.so python profiles/knob_probe.py; done
