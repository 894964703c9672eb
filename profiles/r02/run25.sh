# work regrouping, dealt out (every wave gets one query of each work stratum) vs sorted vs off
python profiles/knob_probe.py "SAGEICP_REGROUP=0" "SAGEICP_REGROUP_MODE=1" "SAGEICP_REGROUP_MODE=0" "SAGEICP_REGROUP_MODE=1"
