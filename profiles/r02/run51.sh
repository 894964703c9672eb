# one scan instead of two for seeded queries (need mask from the seed's bound)
for v in "" variants/occ5.so; do KNOB_CHILD="merged scan lib=$v" KNOB_LIB=$v python profiles/knob_probe.py; done
KNOB_LIB=variants/occ5.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -4
