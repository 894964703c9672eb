# branch-free evaluation + single-exit double step (counted waits in the scan loop)
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4
python profiles/knob_probe.py "" ""
PHASE_LIB=variants/tNN.so PHASE_KIND=NN python profiles/phase_probe.py 1 8
PHASE_LIB=variants/tGN.so PHASE_KIND=GN python profiles/phase_probe.py 1 8
