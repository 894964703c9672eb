for W in 1 2 4; do for g in 8 4 2; do
SAGEICP_GROUP_MAX=$g timeout 300 python profiles/ablate_probe.py w${W}_g${g}:-DSAGE_NN_WAVES=$W 2>&1 | grep k_nn
done; done
