mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ring_geometry or decider" 2>&1 | tail -8
timeout 600 python profiles/knob_ab.py "c2:cold:2:16 c2:cold:3:16" "" 2>&1 | tail -4
