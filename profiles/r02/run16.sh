python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
SPAN_LIB=variants/tNN.so python profiles/span_probe.py 1 20 100
SPAN_LIB=variants/tNN.so python profiles/span_probe.py 1 20 --div 8
