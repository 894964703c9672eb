# work regrouping inside one workgroup's queries (mode 2) vs off
python profiles/knob_probe.py "SAGEICP_REGROUP=0" "SAGEICP_REGROUP_MODE=2" "SAGEICP_REGROUP=0" "SAGEICP_REGROUP_MODE=2"
